"""Static scan of the library's gfx950 assembly for the hazard that broke ilt_dehoog_bwd_kernel in round 5 (DESIGN.md section 0,
item 6): an instruction that only acts on lanes enabled in EXEC -- a spill reload above all (`scratch_load ... ; Folded Reload`,
`v_accvgpr_read` of a spilled value, a plain VMEM / DS load) -- placed where EXEC is known to be EMPTY:

behind a loop that iterates "while any lane is left" (`s_cbranch_execnz <label above>`, a loop back-edge: the fall-through has
EXEC == 0 for EVERY lane, also those that left the loop early), before the instruction that restores EXEC (`s_or_b64 exec, ...`,
`s_mov_b64 exec, ...`, `s_or_saveexec_b64`, ...).  Such an instruction is a no-op: a value the following code expects reloaded
keeps whatever the register held.  ROCm 7.2 produced exactly that for a loop-invariant reload behind a per-lane copy loop.
(NOT a hazard, and only listed with --all: reloads at the target of a forward `s_cbranch_execz` / in the fall-through of a forward
`s_cbranch_execnz` -- the skipped region did not run for any lane, so nothing was clobbered; accumulations there are no-ops by
design.)

    for f in neurallaplacecontrol_amd/csrc/kernels_*.hip; do hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -o /tmp/asm/$(basename ${f%.hip}).s $f; done
    python tools/scan_exec_hazard.py /tmp/asm/*.s

Follows the fall-through and unconditional jumps (`s_branch`, the s_getpc / s_add / s_setpc long-branch idiom) from each point
where EXEC is empty until EXEC is written, a conditional branch or `s_endpgm`; reports EXEC-dependent instructions met on the way.
Scalar instructions, `v_readlane` / `v_writelane` / `v_readfirstlane` (EXEC-independent or harmless) and waits are ignored."""
import re, sys

EXEC_WRITE = re.compile(r"^\s*s_\w+\s+exec\b|^\s*s_\w*saveexec\w*\s")
LABEL = re.compile(r"^(\.L\w+|[A-Za-z_]\w*):")
HARMLESS = re.compile(r"^\s*(s_|;|v_readlane|v_writelane|v_readfirstlane|$)")


def scan(path):
    lines = open(path).read().split("\n")
    labels = {}
    for i, ln in enumerate(lines):
        m = LABEL.match(ln)
        if m:
            labels[m.group(1)] = i
    kernel, out = None, []
    starts = []  # (line index where EXEC == 0 begins, why)
    for i, ln in enumerate(lines):
        m = LABEL.match(ln)
        if m and not m.group(1).startswith(".L"):
            kernel = m.group(1)
        s = ln.strip()
        if s.startswith("s_cbranch_execnz"):
            tgt = s.split()[1]
            back = tgt in labels and labels[tgt] < i
            starts.append((i + 1, kernel, f"fall-through of {'loop back-edge' if back else 'forward'} `{s}` (line {i + 1})", back))
        elif s.startswith("s_cbranch_execz"):
            tgt = s.split()[1]
            if tgt in labels:
                starts.append((labels[tgt] + 1, kernel, f"target of `{s}` (line {i + 1})", False))
    for start, kern, why, hazard in starts:
        i, hops, pending_long = start, 0, None
        while i < len(lines) and hops < 6:
            ln = lines[i]
            s = ln.split(";")[0].strip()
            if not s or LABEL.match(ln):
                i += 1
                continue
            if EXEC_WRITE.match(ln) or s.startswith("s_endpgm"):
                break
            if s.startswith("s_branch"):
                tgt = s.split()[1]
                if tgt not in labels:
                    break
                i, hops = labels[tgt] + 1, hops + 1
                continue
            if s.startswith("s_add_u32") and "-.Lpost_getpc" in s:  # long branch: s_getpc; s_add (label - post); s_addc; s_setpc
                m = re.search(r"\((\.L\w+)-\.Lpost_getpc", s)
                pending_long = m.group(1) if m else None
            if s.startswith("s_setpc_b64"):
                if pending_long in labels:
                    i, hops, pending_long = labels[pending_long] + 1, hops + 1, None
                    continue
                break
            if s.startswith("s_cbranch"):
                break  # a conditional branch on something else: both paths would have to be followed; stop (conservative miss)
            if not HARMLESS.match(s):
                out.append((kern, i + 1, s, why, hazard and ("Folded Reload" in ln or s.startswith("v_accvgpr_read"))))
            i += 1
    # benign: reload - modify - spill of the SAME slot inside one EXEC == 0 region (an accumulation for the active lanes: none)
    def slot(ins):
        m = re.search(r"\boff(?: offset:(\d+))?\s*$", ins)  # scratch_load vX, off, off [offset:N] / scratch_store off, vX, off [offset:N]
        return (m.group(1) or "0") if m else None

    spilled = {(k, w, slot(ins)) for k, _, ins, w, _ in out if ins.startswith("scratch_store")}
    return [(k, ln, ins, w, rel and not (ins.startswith("scratch_load") and (k, w, slot(ins)) in spilled)) for k, ln, ins, w, rel in out]


if __name__ == "__main__":
    total = 0
    for p in sys.argv[1:]:
        hits = scan(p)
        reloads = [h for h in hits if h[4]]
        total += len(reloads)
        print(f"{p}: {len(reloads)} spill reload(s) behind an EXEC-empty loop exit ({len(hits)} EXEC-dependent instructions in all EXEC == 0 regions)")
        for kern, line, ins, why, rel in hits:
            if rel or "--all" in sys.argv:
                print(f"   {'RELOAD ' if rel else ''}{kern} line {line}: {ins}   <- {why}")
    sys.exit(1 if total else 0)
