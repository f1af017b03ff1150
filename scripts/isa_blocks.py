"""Instruction mix per basic block of one kernel in hipcc's -S output (VALU / SALU / LDS / VMEM / MFMA / branch counts).

    hipcc -S --cuda-device-only -O3 --offload-arch=gfx950 ... -o k.s file.hip
    python scripts/isa_blocks.py k.s <kernel-name-substring> [min_instructions]

Prints, per label-delimited block, the line range and the counts, plus the branch targets -- enough to find the loops
of a kernel that is VALU-issue bound (a wave64 VALU instruction occupies a 16-lane SIMD for 4 cycles) and to see what a
loop iteration costs."""
import re
import sys


def kind(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("v_cmpx", "v_")):
        return "valu"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
        return "br"
    if op.startswith(("s_waitcnt", "s_barrier", "s_nop", "s_sleep")):
        return "wait"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    path, name = sys.argv[1], sys.argv[2]
    minn = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^[_A-Za-z0-9]+:", l) and name in l)
    blocks = []
    cur = {"label": "entry", "line": start + 1, "n": {}, "tgt": [], "src": set()}
    for i in range(start + 1, len(lines)):
        l = lines[i]
        if l.startswith(".Lfunc_end"):
            break
        t = l.split(";")[0].strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            blocks.append(cur)
            cur = {"label": m.group(1), "line": i + 1, "n": {}, "tgt": [], "src": set()}
            continue
        if not t or t.startswith("."):
            if t.startswith(".loc"):
                p = t.split()
                if len(p) >= 3:
                    cur["src"].add(int(p[2]))
            continue
        op = t.split()[0]
        k = kind(op)
        cur["n"][k] = cur["n"].get(k, 0) + 1
        if k == "br":
            mm = re.search(r"(\.LBB\d+_\d+)", t)
            if mm:
                cur["tgt"].append(mm.group(1))
    blocks.append(cur)
    tot = {}
    for b in blocks:
        n = b["n"]
        for k, v in n.items():
            tot[k] = tot.get(k, 0) + v
        if sum(n.values()) < minn:
            continue
        src = "%d-%d" % (min(b["src"]), max(b["src"])) if b["src"] else ""
        print("%-12s L%-7d valu %4d salu %4d lds %3d vmem %3d mfma %3d wait %3d  -> %s  %s" % (
            b["label"], b["line"], n.get("valu", 0), n.get("salu", 0) + n.get("smem", 0), n.get("lds", 0), n.get("vmem", 0),
            n.get("mfma", 0), n.get("wait", 0), ",".join(b["tgt"]), src))
    print("total", tot)


if __name__ == "__main__":
    main()
