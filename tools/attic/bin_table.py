#!/usr/bin/env python3
"""Per-bin instruction table of the headline slice kernels from the compiler's assembly (make -C llcomp_amd/csrc asm):
the innermost loops (one iteration = one bin of a run on one slot: unary tail, mantissa tail) and the once-per-sample
slots, every instruction with its class.

    python tools/bin_table.py [llcomp_amd/csrc/slice_kernels.s] > profiles/r03_bin_instruction_table.txt

Classes = issue cost per wavefront-instruction per SIMD at 8 wavefronts per SIMD (tools/ubench/valu_rate3.hip,
profiles/r03_valu_rate3.txt):
  V2  2.4-2.7 cycles: add / sub / and / or / xor / mov, RIGHT shifts (constant or register amount), cndmask on VCC -- with
      VGPR, inline-constant or literal operands
  V4  4.1-4.6 cycles: everything else -- compares, carry ops, 24-bit multiply, SDWA, VOP3, min / max, ffbh, LEFT shifts (even by a
      constant), 64-bit shifts, and any V2 operation that takes a SCALAR register operand
  S   scalar ALU, exec-mask bookkeeping, s_waitcnt, s_nop (one scalar unit per CU: an operation that reads or writes EXEC
      costs ~4 cycles of it per SIMD, s_nop 0.75)
  B   branch (s_cbranch_*, s_branch)
  L   LDS access
  M   global memory access
The encoder's loops are hand-written (llcomp_amd/csrc/enc_rows_asm.hpp): they are found by their labels inside the block."""
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "llcomp_amd/csrc/slice_kernels.s"
V2 = {"v_add_u32_e32", "v_sub_u32_e32", "v_subrev_u32_e32", "v_mov_b32_e32", "v_and_b32_e32", "v_or_b32_e32", "v_xor_b32_e32",
      "v_lshrrev_b32_e32", "v_ashrrev_i32_e32", "v_cndmask_b32_e32", "v_not_b32_e32", "v_mov_b64_e32"}


def klass(text):
    op = text.split()[0]
    if op.startswith(("s_cbranch", "s_branch")):
        return "B"
    if op.startswith("s_"):
        return "S"
    if op.startswith("ds_"):
        return "L"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "M"
    if op.startswith("v_"):
        scalar_operand = re.search(r",\s*s(\d+|\[)", text) is not None
        return "V2" if op in V2 and not scalar_operand else "V4"
    return "?"


def kernel_lines(text, mangled_part):
    lines = text.split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + mangled_part + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))  # (a kernel may hold several s_endpgm)
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
            k = klass(t)
            tot[k] = tot.get(k, 0) + 1
            print(f"   {k:2s}  {t}")
    print("   => " + "  ".join(f"{k}: {v}" for k, v in sorted(tot.items())))


def asm_loops(lines, names):
    """the loops of a hand-written block: from its label up to the branch back to the label"""
    for name, kind in names:
        for i, l in enumerate(lines):
            m = re.match(r"^\s*(" + re.escape(name) + r"\d+):", l)
            if not m:
                continue
            body = []
            for t in lines[i + 1:]:
                t = t.split(";")[0].strip()
                if not t or t.startswith(".p2align"):
                    continue
                if t.endswith(":"):
                    if re.match(r"^\.L(back|skip|patch|rf|mr|xr)", t):
                        continue
                    break
                body.append(t)
                if re.match(r"^s_(branch|cbranch_\w+) " + re.escape(m.group(1)) + r"$", t):
                    break
            print(f"\n-- hand-written loop {m.group(1)}  ({kind}; rare paths are out of line)")
            tot = {}
            for t in body:
                k = klass(t)
                tot[k] = tot.get(k, 0) + 1
                print(f"   {k:2s}  {t}")
            print("   => " + "  ".join(f"{k}: {v}" for k, v in sorted(tot.items())))
            break  # (the block is instantiated more than once: identical text)


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
                k = klass(t)
                n_all[k] = n_all.get(k, 0) + 1
        print("static instruction count of the whole kernel: " + "  ".join(f"{k}: {v}" for k, v in sorted(n_all.items())))
        if "k_encode" in key:
            asm_loops(kernel_lines(text, key), ((".Ltail_", "encoder unary tail: one iteration = one bin on slot 4"),
                                                (".Lman_", "encoder mantissa tail: one iteration = one bin on slot 6")))
        else:
            asm_loops(kernel_lines(text, key), ((".Lu_", "decoder unary tail: one iteration = one bin on slot 4"),
                                                (".Lm_", "decoder mantissa tail: one iteration = one bin on slot 6")))
            print("\n(the compiler's loops below belong to the CHECKED replay path, entered for well under 1 % of the samples)")
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
