"""EXPERIMENT (round 4, not part of the build): a post-pass over the gfx950 assembly of the kernels, run between the
compiler's code generation and the assembler (hipcc --cuda-device-only -S -> this pass -> clang -x assembler -> lld ->
clang-offload-bundler -> hipcc --cuda-host-only -Xclang -fcuda-include-gpubinary: the steps hipcc runs itself).
Result: no kernel moved (profiles/r4_isa_pass_ab.txt) -- the 23-cycle select below is a property of back-to-back
selects in a microbenchmark loop, not of the kernels' instruction streams.  Kept for the record.

What it was meant to fix -- measured on MI355X with tools/bench_micro/valubench.hip (profiles/r4_valubench.txt):

    v_cndmask_b32_e32 d, a, b, vcc      (VOP2 encoding, the condition is the IMPLICIT operand vcc)
        right behind the VALU compare that wrote vcc ........................  ~2 cycles of its SIMD
        vcc written by a scalar instruction, or already read by an earlier
        v_cndmask since the compare (one compare feeding several selects) ....  ~23 cycles
    v_cndmask_b32_e64 d, a, b, vcc      (VOP3 encoding, vcc / an SGPR pair as an explicit operand)
        always ..............................................................  ~4 cycles

The compiler's instruction shrinking turns every select whose condition sits in vcc into the VOP2 form; branch-free code
(one compare, several selects; conditions combined with s_and / s_or) is full of the slow case: 57 of the 110 vcc selects
of k_split, 350 of 469 in k_star_sort_small.  The pass re-encodes as VOP3 every v_cndmask_b32_e32 ..., vcc that is not
the FIRST reader of a vcc value written by a VALU instruction (same operands, same result: only the encoding -- 8 bytes
instead of 4 -- and the operand path change).  Nothing else is touched; an instruction whose source 0 is a literal or an
SGPR (not encodable next to vcc in VOP3 on gfx9) is left alone.
"""
import re
import sys

_INLINE_CONST = re.compile(r"^-?(\d+|0\.5|1\.0|2\.0|4\.0|0x[0-9a-fA-F]+)$")
_VALU_VCC_WRITER = re.compile(
    r"^(v_cmpx?_\w+?_e32\b|v_cmpx?_\w+\s+vcc\b|v_(add|sub|subrev)_co_u32(_e32|_e64)?\s+\w+\s*,\s*vcc\b|"
    r"v_(addc|subb|subbrev)_co_u32(_e32|_e64)?\s+\w+\s*,\s*vcc\b|v_div_scale_\w+\s+[^,]+,\s*vcc\b|v_mad_[iu]64_[iu]32\s+[^,]+,\s*vcc\b)")
_VCC_IMPLICIT_READER = re.compile(r"^(v_(addc|subb|subbrev)_co_u32|v_div_fmas_)")
_SALU_VCC_WRITER = re.compile(r"^s_\w+\s+vcc(_lo|_hi)?\b")


def _is_inline_or_vgpr(tok):
    tok = tok.strip()
    if tok.startswith("v") and (tok[1:].isdigit() or tok.startswith("v[")):
        return True
    if tok in ("0", "1", "-1"):
        return True
    m = _INLINE_CONST.match(tok)
    if not m:
        return False
    try:
        if tok.startswith("0x") or tok.startswith("-0x"):
            return False   # a hexadecimal operand is a 32-bit literal
        v = float(tok)
        return (v == int(v) and -16 <= v <= 64) or tok in ("0.5", "1.0", "2.0", "4.0", "-0.5", "-1.0", "-2.0", "-4.0")
    except ValueError:
        return False


def rewrite(text):
    """-> (new text, number of selects re-encoded, number kept as VOP2, number left alone as not encodable)"""
    out = []
    fresh = False   # vcc holds a value written by a VALU instruction that no select has read yet
    n_re = n_keep = n_skip = 0
    for line in text.split("\n"):
        code = line.split(";", 1)[0].strip()
        if not code or code.startswith(".") and not code.endswith(":"):
            out.append(line)
            continue
        if code.endswith(":"):   # a label: control flow merges here
            fresh = False
            out.append(line)
            continue
        if code.startswith("v_cndmask_b32_e32") and re.search(r",\s*vcc\s*$", code):
            if fresh:
                fresh = False
                n_keep += 1
                out.append(line)
                continue
            ops = code[len("v_cndmask_b32_e32"):].split(",")
            if len(ops) == 4 and _is_inline_or_vgpr(ops[1]):
                out.append(line.replace("v_cndmask_b32_e32", "v_cndmask_b32_e64", 1))
                n_re += 1
            else:
                out.append(line)
                n_skip += 1
            continue
        if _VALU_VCC_WRITER.match(code):
            fresh = True
            if _VCC_IMPLICIT_READER.match(code):
                pass   # (v_addc reads and rewrites vcc: the new value is fresh)
        elif _VCC_IMPLICIT_READER.match(code) or _SALU_VCC_WRITER.match(code):
            fresh = False
        out.append(line)
    return "\n".join(out), n_re, n_keep, n_skip


def main(argv):
    src, dst = argv[1], argv[2]
    mode = argv[3] if len(argv) > 3 else "first"
    text = open(src).read()
    if mode == "all":   # every vcc select as VOP3 (A/B experiment)
        new, n = re.subn(r"v_cndmask_b32_e32(\s+[^,\n]+,\s*(?:v\d+|v\[\d+:\d+\]|-?\d+|-?\d\.\d)\s*,[^,\n]+,\s*vcc\s*(?:;.*)?)$",
                         r"v_cndmask_b32_e64\1", text, flags=re.M)
        print("isa_pass: %d selects re-encoded as VOP3 (all)" % n)
    else:
        new, n_re, n_keep, n_skip = rewrite(text)
        print("isa_pass: %d vcc selects re-encoded as VOP3, %d first readers kept as VOP2, %d not encodable" % (n_re, n_keep, n_skip))
    open(dst, "w").write(new)


if __name__ == "__main__":
    main(sys.argv)
