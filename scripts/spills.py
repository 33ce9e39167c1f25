"""Register / scratch / LDS usage of every kernel in the built objects (from the code-object metadata notes):
python scripts/spills.py [objdir] [name filter]"""
import os, re, subprocess, sys
objdir = sys.argv[1] if len(sys.argv) > 1 else "audiosourcesep_amd/csrc/_obj"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
RO = "/opt/rocm/lib/llvm/bin/llvm-readelf"
for o in sorted(os.listdir(objdir)):
    if not o.endswith(".o"):
        continue
    path = os.path.join(objdir, o)
    # device code object is embedded: extract with clang-offload-bundler
    tmp, fb = "/tmp/_co_%s.co" % o, "/tmp/_fb_%s.bin" % o
    subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fb, path], check=False, capture_output=True)
    subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fb,
                    "--output=" + tmp, "--unbundle"], check=False, capture_output=True)
    if not os.path.exists(tmp):
        continue
    txt = subprocess.run([RO, "--notes", tmp], capture_output=True, text=True).stdout
    for blk in txt.split("- .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
        name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"^void ", "", name).split("(")[0]
        if flt in name:
            print("%-14s %-64s vgpr %4s agpr %4s spill %4s scratch %5s lds %6s" % (o, name[:64], g("vgpr_count"), blk.split()[0], g("vgpr_spill_count"),
                                                                                 g("private_segment_fixed_size"), g("group_segment_fixed_size")))
