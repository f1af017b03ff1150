"""VGPR liveness of one kernel in hipcc's -S output: where the register pressure peaks and what is live there.

    hipcc -S --cuda-device-only -O3 --offload-arch=gfx950 ... -o k.s file.hip
    python scripts/vgpr_liveness.py k.s <kernel-name-substring> [top]

Builds the CFG from the labels / branches of the kernel's assembly, runs the usual backward dataflow over VGPRs
(v0..v255; AGPRs ignored) and prints the instruction with the most live registers plus, for every register live there,
the instruction that defined it and its next use.  Written for kernels that sit at their 128-VGPR cap (2 x 512-thread
workgroups per CU): a spill shows up as scratch_store / scratch_load, this shows which long-lived values crowd the peak.
Approximations: EXEC-masked partial writes count as full definitions; SDWA / DPP destinations with *_PRESERVE are
treated as read-modify-write; inline asm is opaque."""
import re
import sys


def vregs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.append(int(m.group(3)))
    return out


NO_DST = ("s_", "v_cmp", "v_cmpx", "v_readlane", "v_readfirstlane", "ds_write", "ds_store", "global_store",
          "scratch_store", "buffer_store", "flat_store", "s_waitcnt", "s_barrier", "s_nop", "global_atomic_add_f32",
          "ds_add", "ds_max", "ds_min", "ds_or", "ds_and")
RMW = ("v_mfma", "v_fmac", "v_mac", "v_pk_fmac", "v_writelane", "v_dot2c", "v_dot4c")


def parse(path, name):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^[_A-Za-z0-9]+:", l) and name in l)
    body = []
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        body.append(l)
    return body


def main():
    path, name = sys.argv[1], sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    body = parse(path, name)
    ins = []  # (text, defs, uses, label or None, branch targets, falls through)
    labels = {}
    for raw in body:
        l = raw.split(";")[0].rstrip()
        if not l.strip():
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", l.strip())
        if m:
            labels[m.group(1)] = len(ins)
            continue
        t = l.strip()
        if t.startswith("."):
            continue
        parts = t.split(None, 1)
        op = parts[0]
        args = [a.strip() for a in parts[1].split(",")] if len(parts) > 1 else []
        targets, fall = [], True
        if op.startswith("s_cbranch"):
            targets = [args[0]]
        elif op == "s_branch":
            targets, fall = [args[0]], False
        elif op in ("s_endpgm", "s_setpc_b64"):
            fall = False
        nodst = any(op.startswith(p) for p in NO_DST) or "atomic" in op and "rtn" not in op and "glc" not in t and "sc0" not in t
        defs = [] if nodst or not args else vregs(args[0])
        uses = []
        for a in (args if nodst else args[1:]):
            uses += vregs(a)
        if any(op.startswith(p) for p in RMW) or "PRESERVE" in t or "dst_sel:WORD" in t or "dst_sel:BYTE" in t or "op_sel" in t and "v_cvt_pk" not in op:
            uses += defs
        if op.startswith("v_cndmask") or op.startswith("v_mov") or True:
            pass
        ins.append((t, set(defs), set(uses), targets, fall))
    n = len(ins)
    succ = []
    for i, (t, d, u, targets, fall) in enumerate(ins):
        s = [labels[x] for x in targets if x in labels]
        if fall and i + 1 < n:
            s.append(i + 1)
        succ.append(s)
    live_in = [set() for _ in range(n)]
    live_out = [set() for _ in range(n)]
    changed = True
    while changed:
        changed = False
        for i in range(n - 1, -1, -1):
            out = set()
            for s in succ[i]:
                out |= live_in[s]
            inn = ins[i][2] | (out - ins[i][1])
            if out != live_out[i] or inn != live_in[i]:
                live_out[i], live_in[i] = out, inn
                changed = True
    order = sorted(range(n), key=lambda i: -len(live_out[i]))
    print(f"{n} instructions; peak live VGPRs {len(live_out[order[0]])}")
    seen = []
    for i in order:
        if all(abs(i - j) > 40 for j in seen):
            seen.append(i)
        if len(seen) >= top:
            break
    for i in seen:
        print(f"\n== instruction {i}: {ins[i][0]}   live-out {len(live_out[i])}")
        for r in sorted(live_out[i]):
            d = next((j for j in range(i, -1, -1) if r in ins[j][1]), None)
            u = next((j for j in range(i + 1, n) if r in ins[j][2]), None)
            print(f"  v{r:<3d} def@{d}: {ins[d][0][:60] if d is not None else '(entry)':60s} next use@{u}: "
                  f"{ins[u][0][:50] if u is not None else '(loop-carried / none ahead)'}")


if __name__ == "__main__":
    main()
