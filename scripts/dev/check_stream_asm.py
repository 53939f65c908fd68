"""Desk check of attention_stream.hip's generated code (hipcc -save-temps .s): the inline-asm S MFMAs are invisible to the compiler's
hazard recogniser, so this script verifies on the instruction stream what the recogniser would have enforced:
  * no instruction other than the next MFMA of the same accumulation chain touches an asm MFMA's destination registers within WAIT
    wait states behind it (an 8-pass MFMA needs 11 before a VALU / memory read of its result; every instruction counts as one wait
    state, s_nop N as N + 1, another 8-pass MFMA as 8);
  * an asm MFMA's destination never overlaps its A / B operands.
Scans in layout order and follows fall-through only; a branch inside the window is reported (the window then has to be argued by hand).
usage: python scripts/dev/check_stream_asm.py file.s [kernel_substring]"""
import re
import sys

WAIT = 12
path = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else "attn_stream_kernel"
text = open(path).read().split("\n")


def regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def aregs(tok):
    out = set()
    for m in re.finditer(r"\ba\[(\d+):(\d+)\]|\ba(\d+)\b", tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


problems, n_asm, kernels = [], 0, 0
i = 0
while i < len(text):
    if re.match(r"^_Z\w*%s\w*:" % key, text[i]):
        kernels += 1
        name = text[i][:-1]
        j = i + 1
        ins = []                      # (line no, text, in_asm)
        in_asm = False
        while j < len(text) and not text[j].strip().startswith(".end_amdhsa_kernel") and not text[j].startswith("\t.section"):
            t = text[j].strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
            elif t.startswith(";;#ASMEND"):
                in_asm = False
            elif t and not t.startswith((";", ".", "//")) and not re.match(r"^\.?\w+:", t):
                ins.append((j + 1, t, in_asm))
            j += 1
        for a, (ln, t, asm) in enumerate(ins):
            if not (asm and t.startswith("v_mfma")):
                continue
            n_asm += 1
            ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
            dst, srca, srcb, srcc = regs(ops[0]), regs(ops[1]), regs(ops[2]), regs(ops[3]) if len(ops) > 3 else set()
            if dst & (srca | srcb):
                problems.append((name[-40:], ln, "dst overlaps A/B", t))
            if srcc and srcc != dst and (srcc & dst):
                problems.append((name[-40:], ln, "dst partially overlaps C", t))
            waited = 0
            for b in range(a + 1, len(ins)):
                ln2, t2, asm2 = ins[b]
                if waited >= WAIT:
                    break
                if t2.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                    problems.append((name[-40:], ln, f"branch {b - a} instructions behind an asm MFMA", t2))
                    break
                ops2 = [o.strip() for o in t2.split(None, 1)[1].split(",")] if " " in t2 else []
                touched = set().union(*[regs(o) for o in ops2]) if ops2 else set()
                if not (touched & dst):
                    m = re.match(r"s_nop (\d+)", t2)
                    # s_nop N = N + 1 wait states; another MFMA = 8 (it cannot issue before the matrix pipe has spent the 8 passes of the
                    # one before it)
                    waited += int(m.group(1)) + 1 if m else 8 if t2.startswith("v_mfma") else 1
                    continue
                # the next MFMA of the same chain: asm, same dst, srcC == dst
                if asm2 and t2.startswith("v_mfma") and regs(ops2[0]) == dst and len(ops2) > 3 and regs(ops2[3]) == dst and not (dst & (regs(ops2[1]) | regs(ops2[2]))):
                    break             # the chain continues: the window restarts at that MFMA
                problems.append((name[-40:], ln, f"result touched {b - a} instructions later (line {ln2})", t2))
                break
        i = j
    else:
        i += 1
# ---- second audit: the in-place Q loads.  `global_load_dwordx4 a[..]` from inline asm writes accumulator registers asynchronously; the
# compiler believes them written at once.  Between such a load and the next asm `s_waitcnt vmcnt(0)` (layout order) no instruction may
# name those registers (a compiler copy / spill there would move bytes that have not landed).
# ---- third audit: an asm MFMA's AGPR operand (the Q fragment) must not be written within the three instructions in front of it
# (v_accvgpr_write / a load -> MFMA read needs wait states the compiler cannot place for an asm; it also means the fragments do not
# live in the accumulator file, as the kernel's register budget assumes)
flat = []
in_asm = False
for ln, raw in enumerate(text, 1):
    t = raw.strip()
    if t.startswith(";;#ASMSTART"):
        in_asm = True
        continue
    if t.startswith(";;#ASMEND"):
        in_asm = False
        continue
    if not t or t.startswith((";", ".", "//")) or re.match(r"^\.?\w+:", t):
        continue
    flat.append((ln, t.split(";")[0].strip(), in_asm))
n_fresh = 0
for k, (ln, t, a) in enumerate(flat):
    if a and t.startswith("v_mfma"):
        src = aregs(t.split(",")[2])
        states = 0                                            # wait states between the write and the MFMA (s_nop N = N + 1, anything else 1)
        for j in range(k - 1, max(-1, k - 4), -1):
            ln2, t2, _ = flat[j]
            if t2.startswith(("v_accvgpr_write", "scratch_load", "global_load")) and (aregs(t2.split(",")[0]) & src):
                n_fresh += 1
                if states < 4:
                    problems.append(("q-operand", ln, f"AGPR operand written {k - j} instructions earlier (line {ln2}) with {states} wait states", t))
                break
            m = re.match(r"s_nop (\d+)", t2)
            states += int(m.group(1)) + 1 if m else 1

n_qloads = 0
pending = {}                    # register -> line of the load
in_asm = False
for ln, raw in enumerate(text, 1):
    t = raw.strip()
    if t.startswith(";;#ASMSTART"):
        in_asm = True
        continue
    if t.startswith(";;#ASMEND"):
        in_asm = False
        continue
    if not t or t.startswith((";", ".", "//")) or re.match(r"^\.?\w+:", t):
        continue
    body = t.split(";")[0]
    if in_asm and body.startswith("global_load_dwordx4 a["):
        n_qloads += 1
        for rg in aregs(body.split(",")[0]):
            pending[rg] = ln
        continue
    if in_asm and body.startswith("s_waitcnt vmcnt(0)"):
        pending = {}
        continue
    if pending:
        hit = aregs(body) & set(pending)
        if hit:
            problems.append(("q-load", ln, f"a{min(hit)} named before its load (line {pending[min(hit)]}) was waited for", body))
            for rg in hit:
                pending.pop(rg, None)
print(f"{kernels} kernels, {n_asm} asm MFMAs checked, {n_fresh} of them with a Q fragment copied in front (it lives in arch VGPRs there), "
      f"{n_qloads} in-place Q loads checked, {len(problems)} problems")
for p in problems[:40]:
    print(p)
sys.exit(1 if problems else 0)
