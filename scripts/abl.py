"""Diagnostic: time the level-0 f16x3 kernel with parts compiled out (GLOWK_ABL mask; results are wrong for ABL != 0)."""
import os, sys, subprocess
for abl in (0, 1, 2, 3, 6, 7):
    env = dict(os.environ, GLOWK_ABL=str(abl), GLOWK_PREC="1")
    out = subprocess.run([sys.executable, "scripts/time_pieces.py"], env=env, capture_output=True, text=True, timeout=200).stdout
    line = [l for l in out.splitlines() if l.startswith("N=1024")]
    print("ABL", abl, line[0] if line else out[-300:], flush=True)
