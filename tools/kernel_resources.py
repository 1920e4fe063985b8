"""Compile the device side of libocn_mi355x to gfx950 assembly (no GPU needed) and print, for every kernel whose name contains the
given substring: VGPRs, SGPRs, LDS, scratch, code bytes and the count of selected instruction mnemonics.
python tools/kernel_resources.py [substring] [extra hipcc flags ...]"""
import os, re, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1] if len(sys.argv) > 1 else "role_tendency"
extra = sys.argv[2:]
out = "/tmp/ocn_api_gfx950.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                       "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-w", "--cuda-device-only", "-S", "-o", out,
                       os.path.join(ROOT, "oldoceananigans.jl_amd", "csrc", "ocn_api.hip")] + extra)
s = open(out).read()
meta = {}
for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", s, re.S):
    g = lambda k: re.search(k + r"\s+(\S+)", m.group(2)).group(1)
    meta[m.group(1)] = (g("next_free_vgpr"), g("next_free_sgpr"), g("group_segment_fixed_size"), g("private_segment_fixed_size"))
for name, (v, sg, lds, scr) in meta.items():
    if pat not in name:
        continue
    body = re.search(r"^" + re.escape(name) + r":.*?\n(.*?)\n\.Lfunc_end", s, re.S | re.M)
    ops = collections.Counter()
    nbytes = 0
    if body:
        for line in body.group(1).splitlines():
            t = line.strip().split()
            if t and re.match(r"^[vsdb][a-z_0-9]+$", t[0]) and not t[0].endswith(":"):
                ops[t[0]] += 1
    f64 = sum(n for o, n in ops.items() if o.endswith("_f64") and o.startswith("v_"))
    sel = ops["v_cndmask_b32_e32"] + ops["v_cndmask_b32_e64"]
    mov = sum(n for o, n in ops.items() if o.startswith("v_mov") or o.startswith("v_accvgpr"))
    ld = sum(n for o, n in ops.items() if o.startswith("buffer_load") or o.startswith("global_load"))
    print(f"{name[:90]}\n   vgpr {v} sgpr {sg} lds {lds} scratch {scr} | static instr {sum(ops.values())}: v_*_f64 {f64}, v_cndmask {sel}, "
          f"v_mov {mov}, loads {ld}, ds {sum(n for o, n in ops.items() if o.startswith('ds_'))}, s_waitcnt {ops['s_waitcnt']}, "
          f"branches {sum(n for o, n in ops.items() if o.startswith('s_cbranch'))}")
