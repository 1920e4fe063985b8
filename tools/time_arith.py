"""role tendency kernel at 256^3 in both arithmetic modes (option "arithmetic" 0 | 1), twice each, alternating: plain launch and in-step
average (GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for mode in ("0", "1", "0", "1"):
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "time_roles.py")], env=dict(os.environ, OCN_ARITHMETIC=mode), cwd=ROOT, check=True)
