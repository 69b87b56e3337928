#!/usr/bin/env python3
"""Per-bin instruction table of the headline slice kernels from the compiler's assembly (make -C llcomp_amd/csrc asm):
the innermost loops (one iteration = one bin of a run on one slot: unary tail, mantissa tail) and the once-per-sample
slots, every instruction with its class.

    python tools/bin_table.py [llcomp_amd/csrc/slice_kernels.s] > profiles/r03_bin_instruction_table.txt

Classes and what one more instruction of a class per bin costs (measured, profiles/r03_sensitivity.jsonl, encoder alone on
the GPU, 1010 cycles per wavefront-sample at 8 wavefronts per SIMD):
  V2  simple vector op (VOP1/VOP2 e32: add, sub, mov, and, or, shift by a constant, cndmask on VCC)          ~1.0 cycle
  V4  VOP3 / SDWA / VOPC / carry ops / 24-bit multiply (two dwords of encoding, or a second pass)              ~3.1 cycles
  S   scalar ALU, exec-mask bookkeeping, s_waitcnt, s_nop                                                      ~1.4 cycles
  B   branch (s_cbranch_*, s_branch)                                                                           (with S)
  L   LDS access                                                                                               ~1.7 cycles (byte store)
  M   global memory access"""
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "llcomp_amd/csrc/slice_kernels.s"
V2 = {"v_add_u32_e32", "v_sub_u32_e32", "v_subrev_u32_e32", "v_mov_b32_e32", "v_and_b32_e32", "v_or_b32_e32", "v_xor_b32_e32", "v_lshlrev_b32_e32",
      "v_lshrrev_b32_e32", "v_ashrrev_i32_e32", "v_cndmask_b32_e32", "v_max_i32_e32", "v_min_u32_e32", "v_max_u32_e32", "v_min_i32_e32", "v_not_b32_e32",
      "v_add_u16_e32", "v_mov_b64_e32"}


def klass(op):
    if op.startswith(("s_cbranch", "s_branch")):
        return "B"
    if op.startswith("s_"):
        return "S"
    if op.startswith("ds_"):
        return "L"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "M"
    if op.startswith("v_"):
        return "V2" if op in V2 else "V4"
    return "?"


def kernel_lines(text, mangled_part):
    lines = text.split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + mangled_part + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    return lines[start:end + 1]


def blocks(lines):
    """[(label, comment incl. the continuation lines under a loop header, [instructions])]"""
    out, cur = [], ["entry", "", []]
    for l in lines:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l) or re.match(r"^; %bb\.(\d+):\s*(;.*)?$", l)
        if m:
            out.append(tuple(cur))
            cur = [m.group(1), m.group(2) or "", []]
            continue
        t = l.strip()
        if t.startswith(";") and not cur[2] and ("Loop Header" in t or "Child Loop" in t):
            cur[1] += " " + t
            continue
        if not t or t.startswith((";", ".")):
            continue
        cur[2].append(t.split(";")[0].strip())
    out.append(tuple(cur))
    return out


def loops(bl):
    """the depth-2 loops (one iteration = one bin of a run): header block + every block that says it belongs to that header"""
    res = []
    for h, (lab, com, ins) in enumerate(bl):
        if "Loop Header: Depth=2" not in com:
            continue
        name = lab[2:] if lab.startswith(".L") else lab
        body = [h] + [j for j, (l2, c2, i2) in enumerate(bl) if f"Header={name} Depth=2" in c2]
        res.append((lab, sorted(body)))
    return res


def show(title, bl, idxs, skip_deeper=True):
    print(f"\n-- {title}")
    tot = {}
    for j in idxs:
        lab, com, ins = bl[j]
        if skip_deeper and "Depth=3" in com and "Child Loop" not in com:
            continue
        for t in ins:
            op = t.split()[0]
            k = klass(op)
            tot[k] = tot.get(k, 0) + 1
            print(f"   {k:2s}  {t}")
    print("   => " + "  ".join(f"{k}: {v}" for k, v in sorted(tot.items())))


def main():
    text = open(path).read()
    for title, key in (("ENCODER k_encode_slices<1, rows, u16 symbols> (planar one-row slices: the headline)", "k_encode_slicesILi1ELb1EtLb0"),
                       ("DECODER k_decode_slices<1, rows>", "k_decode_slicesILi1ELb1ELb0")):
        print("=" * 120)
        print(title)
        bl = blocks(kernel_lines(text, key))
        ls = loops(bl)
        n_all = {}
        for lab, com, ins in bl:
            for t in ins:
                k = klass(t.split()[0])
                n_all[k] = n_all.get(k, 0) + 1
        print("static instruction count of the whole kernel: " + "  ".join(f"{k}: {v}" for k, v in sorted(n_all.items())))
        for lab, body in ls:
            # the rare carry-propagation loops of the encoder live at depth 3 below these; classify by content
            ops = " ".join(t for j in body for t in bl[j][2])
            if "v_add_co_u32" in ops and "v_mul_u32_u24" in ops or "v_addc_co_u32" in ops:
                kind = "mantissa tail: one iteration = one bin on slot 6"
            elif "v_mul_u32_u24" in ops:
                kind = "unary tail: one iteration = one bin on slot 4"
            else:
                continue
            show(f"loop {lab}  ({kind}; the carry-propagation sub-loop, entered once in ~1000 renormalisations, is left out)", bl, body)


if __name__ == "__main__":
    main()
