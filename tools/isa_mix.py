"""Instruction mix of a kernel's hottest self-loop, from hipcc's gfx950 assembly (developer tool; no GPU needed).

    python tools/isa_mix.py squidpy_amd/csrc/sqgr_ripley.hip 'k_pair_hist_fastILi0E'

Compiles the translation unit to assembly (the flags of squidpy_amd/_build.py), takes the kernel whose mangled name contains
the pattern, splits it into basic blocks at labels and prints the block that branches back to its own label and holds the
most instructions — per instruction class.  bench.py's VALU-mix ceilings for the Ripley / co-occurrence kernels quote these
counts (they are per loop trip: divide by the pairs a trip evaluates)."""
import collections
import re
import subprocess
import sys
import tempfile

src, pat = sys.argv[1], sys.argv[2]
with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", src, "-o", tmp.name],
                          stderr=subprocess.DEVNULL)
    lines = open(tmp.name).read().split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and pat in l]
if not starts:
    sys.exit(f"no kernel matching {pat}")
for st in starts:
    end = st
    while "s_endpgm" not in lines[end]:
        end += 1
    body = lines[st:end]
    blocks, cur, name = [], [], lines[st].split(":")[0]
    for l in body[1:]:
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            blocks.append((name, cur))
            name, cur = m.group(1), []
        elif l.startswith("\t") and not l.strip().startswith((".", ";")):
            cur.append(l.strip())
    blocks.append((name, cur))
    loops = [(n, b) for n, b in blocks if b and any(i.startswith("s_cbranch") and i.split()[-1] == n for i in b)]
    if not loops:
        print(lines[st].split(":")[0], "no self-loop found")
        continue
    n, b = max(loops, key=lambda t: len(t[1]))

    def cls(i: str) -> str:
        op = i.split()[0]
        if op.startswith("v_") and "f64" in op:
            return "valu_f64:" + op
        if op.startswith("v_"):
            return "valu_32"
        if op.startswith("ds_"):
            return "lds:" + op
        if op.startswith(("global_", "buffer_", "flat_")):
            return "vmem"
        if op.startswith("s_load") or op.startswith("s_buffer"):
            return "smem"
        return "salu/other"

    c = collections.Counter(cls(i) for i in b)
    print(lines[st].split(":")[0])
    print(f"  hottest self-loop {n}: {len(b)} instructions")
    for k, v in sorted(c.items()):
        print(f"    {k:40s} {v}")
