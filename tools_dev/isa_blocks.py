"""per-basic-block instruction mix of one kernel in hipcc's ISA (--save-temps): usage isa_blocks.py <file.s> <mangled-name-prefix>"""
import collections, re, sys
t = open(sys.argv[1]).read()
i = t.index("\n" + sys.argv[2]); i = t.index(":", i)
j = t.index(".Lfunc_end", i)
body = t[i:j].split("\n")
bbs, cur, name = [], [], "entry"
for ln in body:
    s = ln.strip()
    if re.match(r"^\.LBB\d+_\d+:", s):
        bbs.append((name, cur)); name = s; cur = []
    elif s and not s.startswith(";") and not s.startswith("."):
        cur.append(s.split()[0])
bbs.append((name, cur))
for n, c in bbs:
    cnt = collections.Counter(c)
    mf = sum(v for k, v in cnt.items() if "mfma" in k)
    if mf or len(c) > 40:
        valu = sum(v for k, v in cnt.items() if k.startswith("v_") and "mfma" not in k)
        print(n, "n", len(c), "mfma", mf, "valu", valu, "ds", sum(v for k, v in cnt.items() if k.startswith("ds_")), "glob", sum(v for k, v in cnt.items() if k.startswith("global")),
              "salu", sum(v for k, v in cnt.items() if k.startswith("s_")), "nop", cnt.get("s_nop", 0), "wait", cnt.get("s_waitcnt", 0))
        if len(sys.argv) > 3:
            print("   ", dict(cnt.most_common(40)))
