"""The C ABI never aborts ("C ABI returns int status ... never abort"): every entry point of include/ocn_mi355x.h that takes
a pointer is called with all-NULL / all-zero arguments on the GPU box, in a child process (a crash would take the test runner down);
each call must return -- with a non-zero status where an object or an array is required -- and leave a message in ocn_last_error()."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes as C, json, sys
sys.path.insert(0, %r)
from oldoceananigans_jl_amd import _lib
L = _lib.lib()
assert L.ocn_init(0) == 0
out = {}
for name, (restype, argtypes) in sorted(_lib.SYMBOLS.items()):
    if restype is not C.c_int or not any(hasattr(a, "contents") or a is C.c_void_p or a is C.c_char_p for a in argtypes):
        continue
    if name in ("ocn_init",):
        continue
    args = []
    for a in argtypes:
        if a in (C.c_int, C.c_size_t, C.c_long, C.c_int64):
            args.append(0)
        elif a is C.c_double:
            args.append(0.0)
        else:
            args.append(None)
    rc = getattr(L, name)(*args)
    msg = L.ocn_last_error()
    out[name] = [int(rc), (msg or b"").decode(errors="replace")[:80]]
    print(name, rc, flush=True)
print("RESULT " + json.dumps(out))
"""


def test_every_entry_point_returns_on_null_arguments():
    env = dict(os.environ, OCN_TEST_NO_TORCH="1")
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.returncode, r.stdout[-600:], r.stderr[-1200:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    out = json.loads(line[len("RESULT "):])
    assert len(out) > 80
    # NULL is a valid argument only for releasing / waiting on nothing, for process-wide settings and for a copy of zero bytes
    may_succeed = ("destroy", "free", "close", "set_option", "set_stream", "own_stream", "wait", "debug", "memcpy", "memset")
    bad = {n: v for n, v in out.items() if v[0] == 0 and not any(t in n for t in may_succeed)}
    assert not bad, bad
    assert all(v[1] for n, v in out.items() if v[0] != 0), "a failing call left no message"
