import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t = time.time()
import torch
print("import torch", time.time() - t, torch.__version__, flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "cuda":
    print("torch.cuda.is_available", torch.cuda.is_available(), flush=True)
    x = torch.ones(4, device="cuda"); print(x.sum().item())
import __graft_entry__ as g
g.smoke()
os.system(f"grep -E 'amdhip|hipfft|rocfft|rccl|hsa-runtime' /proc/{os.getpid()}/maps | awk '{{print $6}}' | sort -u")
