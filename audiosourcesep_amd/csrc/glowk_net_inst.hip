// One translation unit per level shape: the explicit instantiation of launch_net_t<CI, NF> pulls in every coupling-network
// kernel instance of that shape (fp32 and fp16x3, both tilings, 2 and 4 passes, split or not, three modes).
// Compiled by __graft_entry__.build() with -DGLOWK_INST_CI=<ci> -DGLOWK_INST_NF=<nf>.
#ifndef GLOWK_INST_CI   // (a bare "hipcc -c" of this file still compiles: the level-1 shape of the benchmark config)
#define GLOWK_INST_CI 2
#define GLOWK_INST_NF 16
#endif
#include "glowk_launch.h"

namespace glowk_detail {
template int launch_net_t<GLOWK_INST_CI, GLOWK_INST_NF>(const NetArgs&, int, hipStream_t, bool);
}
