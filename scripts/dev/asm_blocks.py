"""Summarise the basic blocks of one kernel in a hipcc -save-temps .s file: instruction mix per block (MFMA, LDS reads, v_exp,
scratch traffic, AGPR copies, s_nop, waits) - the desk check of a kernel's hot loops when no GPU is at hand.
usage: python scripts/dev/asm_blocks.py file.s kernel_substring [min_mfma]"""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lines = open(path).read().split("\n")
start = next(i for i, ln in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % key, ln))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".end_amdhsa_kernel") or lines[i].startswith("\t.section") and i > start + 5)
blocks, cur = [], ["entry", collections.Counter(), 0]
cats = [("mfma", r"v_mfma"), ("ds_read", r"ds_read"), ("ds_write", r"ds_write"), ("exp", r"v_exp"), ("scr_ld", r"scratch_load"), ("scr_st", r"scratch_store"),
        ("acc_rd", r"v_accvgpr_read"), ("acc_wr", r"v_accvgpr_write"), ("nop", r"s_nop"), ("wait", r"s_waitcnt"), ("pk_add", r"v_pk_add_f32"),
        ("add", r"v_add_f32"), ("cvt", r"v_cvt_pk_bf16"), ("glds", r"global_load_lds|buffer_load.*lds"), ("gload", r"global_load_dword"), ("gstore", r"global_store"),
        ("mov", r"v_mov_b32|v_mov_b64"), ("barrier", r"s_barrier"), ("branch", r"s_cbranch|s_branch"), ("cndmask", r"v_cndmask"), ("valu", r"^\s*v_")]
for ln in lines[start:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", ln)
    if m:
        blocks.append(cur)
        cur = [m.group(1), collections.Counter(), 0]
        continue
    t = ln.strip()
    if not t or t.startswith((";", ".", "//")):
        continue
    cur[2] += 1
    for name, pat in cats:
        if re.search(pat, t):
            cur[1][name] += 1
blocks.append(cur)
tot = collections.Counter()
for name, c, n in blocks:
    tot.update(c)
    if c["mfma"] >= min_mfma or c["scr_ld"] + c["scr_st"] > 8:
        print(f"{name:12s} n={n:5d} " + " ".join(f"{k}={v}" for k, v in c.items()))
print("TOTAL", dict(tot))
