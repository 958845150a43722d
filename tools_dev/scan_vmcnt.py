"""Low vmcnt waits inside the main loop of the matrix-core kernels (hipcc --save-temps ISA): usage scan_vmcnt.py <file.s> [threshold]
A wave of k_mmq_kq keeps a ring of D blocks of weights in flight (LW loads each); a compiler-inserted `s_waitcnt vmcnt(N)` with N far below (D - 1) * LW in the main loop
drains that ring — a false dependency: a register the allocator reused (the undefined half of a pair, a temporary), or wait-count states merged over an early-exit edge.
The main loop is taken to be the code between the first and the last LDS-DMA (`global_load_lds`) of the kernel; the STG form (four chunks: no DMA) is not scanned.
Prints the kernels that hold such a wait, and the count of kernels scanned; exit code 1 if any."""
import re, sys
t = open(sys.argv[1]).read()
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 12
n, bad = 0, 0
for m in re.finditer(r"\n(_Z\S+):[^\n]*\n(.*?)\n\.Lfunc_end", t, re.S):
    name, body = m.group(1), m.group(2)
    if "k_mmq" not in name or "global_load_lds" not in body:
        continue
    n += 1
    a, b = body.index("global_load_lds"), body.rindex("global_load_lds")
    # the prologue's own drain (vmcnt(0) + barrier) sits between the first DMAs; count from the first barrier on
    a = body.index("s_barrier", a) if "s_barrier" in body[a:b] else a
    lows = [int(w) for w in re.findall(r"s_waitcnt vmcnt\((\d+)\)", body[a:b]) if int(w) < thr]
    if lows:
        bad += 1
        print(f"{name[:100]:100s} {lows[:16]}")
print(f"{n} kernels scanned, {bad} with a vmcnt wait below {thr} in the main loop")
sys.exit(1 if bad else 0)
