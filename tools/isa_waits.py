#!/usr/bin/env python3
"""Lists, per kernel of a .hip source, the order of vector-memory loads (L), stores (S), atomics (A), scratch accesses
(X) and s_waitcnt vmcnt(n) (Wn) in the gfx950 ISA — and flags runs of `L W0` / `L W1`: loads that wait for one another.
A load issued under a condition makes the compiler wait at the join; eight gathers written that way were eight
dependent round trips in the permute's tile kernel.  usage: tools/isa_waits.py sbx_permute.hip [kernel-substring]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "sparsebase_amd", "csrc", sys.argv[1])
want = sys.argv[2] if len(sys.argv) > 2 else ""
asm = "/tmp/isa_waits.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"),
                       "-I", os.path.join(ROOT, "sparsebase_amd", "csrc"), "-S", "--cuda-device-only", "-o", asm, src],
                      stderr=subprocess.DEVNULL)
s = open(asm).read()
for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)\.Lfunc_end", s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if ".amdhsa_kernel " + name not in s or want not in name:
        continue
    seq = []
    for l in body.split("\n"):
        t = l.strip().split()
        if not t: continue
        op = t[0]
        if op.startswith(("global_load", "buffer_load", "flat_load")): seq.append("L")
        elif op.startswith(("global_store", "buffer_store", "flat_store")): seq.append("S")
        elif op.startswith(("global_atomic", "flat_atomic")): seq.append("A")
        elif op.startswith("scratch_"): seq.append("X")
        elif op == "s_waitcnt" and "vmcnt" in l: seq.append("W" + re.search(r"vmcnt\((\d+)\)", l).group(1))
    txt = " ".join(seq)
    chains = re.findall(r"(?:L W[01] ){3,}", txt + " ")
    demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    print(f"{demangled[:110]}\n   loads {seq.count('L')} stores {seq.count('S')} scratch {seq.count('X')} | serialized load chains: "
          f"{[c.count('L') for c in chains] or 'none'}")
    if want: print("   " + txt[:1500])
