"""Audit of the hand-counted weight ring of k_mmq_g2 (mmq_mfma.hip) in the compiler's output: between the asm statement that loads a ring slot and the `s_waitcnt ... ; landed:`
statement that names the slot's registers, no other instruction may read or write those registers (a compiler copy or spill of a register whose load is still in flight would
move garbage).  usage: python3 tools_dev/check_asm_ring.py <mmq_mfma-hip-amdgcn-amd-amdhsa-gfx950.s>   (from hipcc --save-temps)"""
import re, sys

def regs_of(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()

def vregs(line):
    out = set()
    for tok in re.findall(r"v\[\d+:\d+\]|v\d+", line):
        out |= regs_of(tok)
    return out

text = open(sys.argv[1]).read()
bad = total = 0
for km in re.finditer(r"^(_Z8k_mmq_g2\S+):.*?\.Lfunc_end\d+:", text, re.S | re.M):
    lines = km.group(0).split("\n")
    pending = {}                      # register -> line number of the load that targets it
    in_asm = False
    for n, ln in enumerate(lines):
        t = ln.strip()
        if t.startswith(";;#ASMSTART"): in_asm = True; continue
        if t.startswith(";;#ASMEND"): in_asm = False; continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        if in_asm and t.startswith("global_load_dword"):
            dst = regs_of(t.split()[1].rstrip(","))
            for r in dst: pending[r] = n
            total += 1
            continue
        if in_asm and "landed:" in t:
            for r in vregs(t.split("landed:")[1]): pending.pop(r, None)
            continue
        if in_asm and t.startswith("s_waitcnt vmcnt(0)") and "landed" not in t:
            continue
        # a WRITE to such a register, or a copy / store out of it, is a violation; a packed-f32 instruction whose 64-bit source pair merely contains it
        # (op_sel picks the other half) is not
        parts = t.split(None, 1)
        ops = parts[1].split(",") if len(parts) > 1 else []
        touched = regs_of(ops[0].strip()) if ops and not parts[0].startswith(("global_store", "scratch_store", "ds_write", "buffer_store")) else set()
        if parts[0].startswith(("v_mov", "v_accvgpr", "scratch_", "global_store", "buffer_store", "ds_write", "v_readlane", "v_readfirstlane")):
            touched = vregs(t)
        hit = touched & set(pending)
        if hit and not in_asm:
            bad += 1
            print(f"{km.group(1)}: line {n}: `{t}` touches v{sorted(hit)} while its load (line {pending[min(hit)]}) is in flight")
# a spilled ring register would be stored before its load has landed: the hand-counted instances must not spill at all
for m in re.finditer(r"\.name:\s+(_Z8k_mmq_g2\S+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", text):
    if int(m.group(2)):
        bad += 1
        print(f"{m.group(1)}: {m.group(2)} spilled VGPRs")
if total == 0:
    bad += 1
    print("no ring loads found: the audit did not see k_mmq_g2 at all")
print(f"{total} ring loads checked, {bad} violations")
sys.exit(1 if bad else 0)
