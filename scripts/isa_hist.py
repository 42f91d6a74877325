#!/usr/bin/env python
"""Instruction histogram of the kernels in a hipcc -save-temps .s file (per function, and per loop body).

  hipcc --offload-arch=gfx950 -O3 ... -save-temps=obj -c x.hip -o /tmp/t/x.o
  python scripts/isa_hist.py /tmp/t/x-hip-amdgcn-amd-amdhsa-gfx950.s k_wide_single
"""
import re
import sys
from collections import Counter


def main():
    text = open(sys.argv[1]).read()
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    for m in re.finditer(r"\n(_Z[^\n:]*):[^\n]*\n(.*?)\n\s*\.end_amdhsa_kernel", text, re.S):
        name, body = m.group(1), m.group(2)
        if pat not in name:
            continue
        code = body.split(".section")[0]
        ins, labels = [], {}
        for line in code.split("\n"):
            t = line.strip()
            if re.match(r"^\.LBB\d+_\d+:", t):
                labels[t.split(":")[0]] = len(ins)
            elif line.startswith("\t") and t and not t.startswith((".", ";")):
                ins.append(t)
        ops = Counter(i.split()[0] for i in ins)
        kinds = lambda c: {"valu": sum(v for k, v in c.items() if k.startswith("v_")),
            "salu": sum(v for k, v in c.items() if k.startswith("s_")),
                           "vmem": sum(v for k, v in c.items() if k.startswith(("global_", "buffer_", "flat_", "scratch_"))),
                               "lds": sum(v for k, v in c.items() if k.startswith("ds_"))}
        vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", body)
        print(f"{name[:100]}\n  instructions {len(ins)} {kinds(ops)} vgpr {vg.group(1) if vg else '?'}")
        # loops: backward branches
        for k, i in enumerate(ins):
            b = re.match(r"s_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)", i)
            if b:
                tgt = b.group(1) or b.group(2)
                if tgt in labels and labels[tgt] <= k:
                    c = Counter(x.split()[0] for x in ins[labels[tgt]:k + 1])
                    print(f"  loop {tgt}: {k + 1 - labels[tgt]} instructions {kinds(c)}")
        if "-v" in sys.argv:
            print("  ", sorted(ops.items(), key=lambda x: -x[1])[:40])


if __name__ == "__main__":
    main()
