#!/usr/bin/env python3
"""Transcribe the three vector math routines that ATen's CPU kernels call for float tensors on an AVX-512 machine —
Sleef_expf16_u10, Sleef_logf16_u10, Sleef_tanhf16_u10 (SLEEF, Boost Software License 1.0), as compiled into the installed
libtorch_cpu.so — into scalar C, one statement per machine instruction with the instruction's exact IEEE semantics.

Why: the reference's CPU path computes `torch.tanh` (PredictUpdate), `torch.sigmoid` = 1 / (1 + exp(-x)) (conv-LSTM) and
`torch.log` (CDF row index) with these routines; their results differ from libm's in the last bits, and a codec's entropy
decisions depend on those bits.  The generated header (plain C, also valid HIP device code) lets the oracle's C back-end
and the GPU kernels reproduce ATen's values bit for bit.  Build-container tool: it reads the machine code and the
constants of the torch installation it runs under and verifies the result against torch itself.

  python tools/sleef_transcribe.py [OUT.h]      # writes the header (default: the oracle's and the product's copy), then
                                                # checks it against torch on ~10^8 inputs
"""
import ctypes
import os
import pickle
import re
import struct
import subprocess
import sys
import tempfile

import numpy as np
import torch

LIB = os.path.join(os.path.dirname(torch.__file__), "lib", "libtorch_cpu.so")
FUNCS = [("Sleef_expf16_u10avx512f", "pm_sleef_expf"), ("Sleef_logf16_u10avx512f", "pm_sleef_logf"),
         ("Sleef_tanhf16_u10avx512f", "pm_sleef_tanhf")]


def elf_segments(path):
    f = open(path, "rb")
    eh = f.read(64)
    phoff = struct.unpack_from("<Q", eh, 32)[0]
    phentsize, phnum = struct.unpack_from("<HH", eh, 54)
    segs = []
    for i in range(phnum):
        f.seek(phoff + i * phentsize)
        p_type, _, off, vaddr, _, filesz, _, _ = struct.unpack("<IIQQQQQQ", f.read(phentsize))
        if p_type == 1:
            segs.append((vaddr, off, filesz))
    return f, segs


def read_u32(f, segs, va):
    for v, o, s in segs:
        if v <= va < v + s:
            f.seek(o + va - v)
            return struct.unpack("<I", f.read(4))[0]
    raise ValueError(hex(va))


def symbol_addr(name):
    out = subprocess.run(f"nm -D --defined-only {LIB} | grep ' {name}$'", shell=True, capture_output=True, text=True).stdout
    return int(out.split()[0], 16)


def disassemble(addr):
    out = subprocess.run(["objdump", "-d", "--no-show-raw-insn", f"--start-address={hex(addr)}",
                          f"--stop-address={hex(addr + 0x800)}", LIB], capture_output=True, text=True).stdout
    lines = []
    for line in out.splitlines():
        m = re.match(r"\s*[0-9a-f]+:\s+(\S+)\s*(.*)$", line)
        if not m:
            continue
        op, rest = m.group(1), m.group(2)
        tgt = re.search(r"#\s+([0-9a-f]+)\s+<", rest)
        rest = re.sub(r"\s*#.*$", "", rest)
        rest = re.sub(r"\s*<[^>]*>", "", rest).strip()
        lines.append((op, rest, int(tgt.group(1), 16) if tgt else None))
        if op == "ret":
            break
    return lines


def reg(tok):
    m = re.match(r"%[xyz]mm(\d+)$", tok)
    assert m, tok
    return f"z{m.group(1)}"


def transpile(name, cname, f, segs):
    body = []
    regs = set()

    def R(tok):
        r = reg(tok)
        regs.add(r)
        return r

    for op, rest, tgt in disassemble(symbol_addr(name)):
        if op == "ret":
            break
        mask = None
        mm = re.search(r"\{%k(\d)\}", rest)
        if mm:
            mask = f"k{mm.group(1)}"
            rest = rest.replace(mm.group(0), "")
        rest = rest.replace("{1to16}", "")
        ops = [t.strip() for t in rest.split(",")]
        imm = None
        if ops and ops[0].startswith("$"):
            imm = int(ops[0][1:], 16)
            ops = ops[1:]

        def src(tok):
            if "(%rip)" in tok:
                return f"0x{read_u32(f, segs, tgt):08x}u"
            return R(tok)

        d = ops[-1]
        st = None
        if op in ("vbroadcastss", "vpbroadcastd", "vmovaps"):
            st = (R(d), src(ops[0]))
        elif op in ("vxorps", "vpxor") and ops[0] == ops[1]:
            st = (R(d), "0u")
        elif op in ("vaddps", "vsubps", "vmulps", "vdivps"):
            a, b = src(ops[1]), src(ops[0])                       # AT&T: op src2, src1, dst  ->  dst = src1 op src2
            c = {"vaddps": "+", "vsubps": "-", "vmulps": "*", "vdivps": "/"}[op]
            st = (R(d), f"PM_U(PM_F({a}) {c} PM_F({b}))")
        elif re.match(r"vfn?m(add|sub)(132|213|231)ps", op):
            m = re.match(r"vf(n?)m(add|sub)(132|213|231)ps", op)
            neg, kind, order = m.group(1) == "n", m.group(2), m.group(3)
            o1, o2, o3 = R(d), src(ops[1]), src(ops[0])           # Intel operand numbering
            x, y, z = {"132": (o1, o3, o2), "213": (o2, o1, o3), "231": (o2, o3, o1)}[order]     # x*y (+/-) z
            xs = f"-PM_F({x})" if neg else f"PM_F({x})"
            zs = f"-PM_F({z})" if kind == "sub" else f"PM_F({z})"
            st = (o1, f"PM_U(PM_FMA({xs}, PM_F({y}), {zs}))")
        elif op == "vcvtps2dq":
            st = (R(d), f"pm_cvtps2dq({src(ops[0])})")
        elif op == "vcvtdq2ps":
            st = (R(d), f"PM_U((float)(int32_t)({src(ops[0])}))")
        elif op == "vpsrad":
            st = (R(d), f"(uint32_t)((int32_t)({src(ops[0])}) >> {imm})")
        elif op == "vpslld":
            st = (R(d), f"({src(ops[0])} << {imm})")
        elif op in ("vpaddd", "vpsubd"):
            a, b = src(ops[1]), src(ops[0])
            st = (R(d), f"({a} {'+' if op == 'vpaddd' else '-'} {b})")
        elif op == "vpandnd":
            st = (R(d), f"(~{src(ops[1])} & {src(ops[0])})")
        elif op in ("vpandd", "vpxord", "vpord"):
            c = {"vpandd": "&", "vpxord": "^", "vpord": "|"}[op]
            st = (R(d), f"({src(ops[1])} {c} {src(ops[0])})")
        elif op == "vpternlogd" and imm == 0xff:
            st = (R(d), "0xffffffffu")
        elif op.startswith("vcmp"):
            pred = op[4:-2]
            a, b = src(ops[1]), src(ops[0])
            expr = {"lt_oq": f"PM_F({a}) < PM_F({b})", "gt_oq": f"PM_F({a}) > PM_F({b})", "eq": f"PM_F({a}) == PM_F({b})",
                    "neq": f"!(PM_F({a}) == PM_F({b}))"}[pred]
            body.append(f"    const int k{d[2:]}_{len(body)} = {expr}; k{d[2:]} = k{d[2:]}_{len(body)};")
            continue
        elif op == "korw":
            body.append(f"    k{d[2:]} = k{ops[1][2:]} | k{ops[0][2:]};")
            continue
        elif op == "vgetmantps":
            assert imm == 0xb
            st = (R(d), f"pm_getmant_075_15({src(ops[0])})")
        elif op == "vgetexpps":
            st = (R(d), f"pm_getexp({src(ops[0])})")
        elif op == "vfixupimmps":
            assert imm == 0
            st = (R(d), f"pm_fixupimm({R(d)}, {src(ops[1])}, {src(ops[0])})")
        else:
            raise NotImplementedError(f"{op} {rest}")
        dst, expr = st
        body.append(f"    if ({mask}) {dst} = {expr};" if mask else f"    {dst} = {expr};")
    decl = ", ".join(f"{r} = 0" for r in sorted(regs - {"z0"}, key=lambda s: int(s[1:])))
    return (f"PM_SLEEF_FN float {cname}(float x) {{\n    uint32_t z0 = PM_U(x), {decl};\n    int k0 = 0, k1 = 0;\n    (void)k0;\n"
            + "\n".join(body) + "\n    return PM_F(z0);\n}\n")


HEADER = '''/* GENERATED by tools/sleef_transcribe.py — do not edit.
 *
 * Scalar restatement, one statement per machine instruction, of the three SLEEF (Boost Software License 1.0) routines that
 * ATen's vectorised CPU kernels call for float tensors on an AVX-512 machine, as compiled into libtorch_cpu.so %s:
 * Sleef_expf16_u10 (torch.exp, torch.sigmoid = 1 / (1 + exp(-x))), Sleef_logf16_u10 (torch.log), Sleef_tanhf16_u10
 * (torch.tanh).  Every operation is a single IEEE binary32 operation (fmaf where the machine code fuses), so the values
 * equal ATen's bit for bit; verified by the generator against torch on ~10^8 inputs per function including every special.
 * (ATen evaluates only whole 16-float vectors this way: the last numel %% 32 elements of each thread's slice of a tensor go
 * through the scalar lambda, i.e. glibc's expf, which differs from SLEEF in the last bit on a few inputs per million.  The
 * gate tensors of the path have 32 channels per plane, so their slices hold whole vectors only.)
 * Valid as C (oracle) and as HIP device code (product): define PM_SLEEF_FN before including.
 */
#ifndef PM_SLEEF_F32_H
#define PM_SLEEF_F32_H
#include <stdint.h>
#ifndef PM_SLEEF_FN
#define PM_SLEEF_FN static inline
#endif
#if defined(__HIPCC__)
#define PM_U(f) __float_as_uint(f)
#define PM_F(u) __uint_as_float(u)
#define PM_FMA(a, b, c) __builtin_fmaf((a), (b), (c))
#define PM_RINT(a) __builtin_rintf(a)
#else
#include <math.h>
#include <string.h>
static inline uint32_t PM_U(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float PM_F(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
#define PM_FMA(a, b, c) fmaf((a), (b), (c))
#define PM_RINT(a) rintf(a)
#endif

/* vcvtps2dq: round to nearest even; out of range / NaN -> 0x80000000 */
PM_SLEEF_FN uint32_t pm_cvtps2dq(uint32_t u) {
    const float f = PM_F(u);
    if (!(f > -2147483648.0f && f < 2147483648.0f)) return 0x80000000u;
    return (uint32_t)(int32_t)PM_RINT(f);
}
/* vgetexpps: floor(log2|x|) as a float; denormals are normalised; 0 -> -inf, inf -> +inf, NaN -> NaN */
PM_SLEEF_FN uint32_t pm_getexp(uint32_t u) {
    const uint32_t a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return u | 0x00400000u;
    if (a == 0x7f800000u) return 0x7f800000u;
    if (a == 0) return 0xff800000u;
    int e = (int)(a >> 23) - 127;
    if ((a >> 23) == 0) {            /* denormal: normalise */
        uint32_t m = a;
        e = -126;
        while (!(m & 0x00800000u)) { m <<= 1; --e; }
    }
    return PM_U((float)e);
}
/* vgetmantps imm 0xb: mantissa normalised to [0.75, 1.5), negative inputs -> QNaN */
PM_SLEEF_FN uint32_t pm_getmant_075_15(uint32_t u) {
    const uint32_t a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return u | 0x00400000u;
    if ((u >> 31) && a != 0) return 0xffc00000u;
    if (a == 0x7f800000u) return 0x3f800000u;
    if (a == 0) return 0x3f800000u;
    uint32_t m = a;
    if ((a >> 23) == 0) { while (!(m & 0x00800000u)) m <<= 1; }
    m &= 0x007fffffu;
    /* [1, 2) -> [0.75, 1.5): mantissas of at least 1.5 go to [0.75, 1) */
    return (m >= 0x00400000u) ? (0x3f000000u | m) : (0x3f800000u | m);
}
/* vfixupimmps imm 0: table nibble per class of src (QNaN, SNaN, zero, one, -inf, +inf, negative, positive) */
PM_SLEEF_FN uint32_t pm_fixupimm(uint32_t dst, uint32_t src, uint32_t table) {
    const uint32_t a = src & 0x7fffffffu;
    int tok;
    if (a > 0x7f800000u) tok = (src & 0x00400000u) ? 0 : 1;
    else if (a == 0) tok = 2;                       /* (denormals are not flushed: MXCSR.DAZ = 0) */
    else if (src == 0x3f800000u) tok = 3;
    else if (src == 0xff800000u) tok = 4;
    else if (src == 0x7f800000u) tok = 5;
    else tok = (src >> 31) ? 6 : 7;
    switch ((table >> (4 * tok)) & 15u) {
    case 0: return dst;
    case 1: return src;
    case 2: return src | 0x00400000u;
    case 3: return 0xffc00000u;
    case 4: return 0xff800000u;
    case 5: return 0x7f800000u;
    case 6: return (src & 0x80000000u) | 0x7f800000u;
    case 7: return 0x80000000u;
    case 8: return 0u;
    case 9: return 0xbf800000u;
    case 10: return 0x3f800000u;
    case 11: return 0x3f000000u;
    case 12: return 0x42b40000u;
    case 13: return 0x3fc90fdbu;
    case 14: return 0x7f7fffffu;
    default: return 0xff7fffffu;
    }
}

'''


def generate(path):
    f, segs = elf_segments(LIB)
    src = HEADER % torch.__version__
    for name, cname in FUNCS:
        src += transpile(name, cname, f, segs) + "\n"
    src += ("/* torch.sigmoid on float CPU tensors: 0 - x, exp, 1 + ., reciprocal (one IEEE division) */\n"
            "PM_SLEEF_FN float pm_aten_sigmoidf(float x) { return 1.0f / (1.0f + pm_sleef_expf(0.0f - x)); }\n\n#endif\n")
    open(path, "w").write(src)
    return src


def verify(header):
    td = tempfile.mkdtemp()
    c = os.path.join(td, "t.c")
    open(c, "w").write(f'#include "{os.path.abspath(header)}"\n'
                       "void run(int which, const float *x, float *y, long n) {\n"
                       "  for (long i = 0; i < n; ++i) y[i] = which == 0 ? pm_sleef_expf(x[i]) : which == 1 ? pm_sleef_logf(x[i])\n"
                       "                                   : which == 2 ? pm_sleef_tanhf(x[i]) : pm_aten_sigmoidf(x[i]);\n}\n")
    so = os.path.join(td, "t.so")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC", "-o", so, c, "-lm"])
    L = ctypes.CDLL(so)
    fp = np.ctypeslib.ndpointer(np.float32, flags="C")
    L.run.argtypes = [ctypes.c_int, fp, fp, ctypes.c_long]
    rng = np.random.default_rng(0)
    ok = True
    specials = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1.1754944e-38, 3.4e38, -3.4e38, 88.7,
                         -88.7, 100.0, 100.00001, 104.0, -104.0, -104.00001, 8.664339, 8.66434, 8.664341, 0.5, 1e-5, 0.01],
                        np.float32)
    for which, (tname, fn) in enumerate([("exp", torch.exp), ("log", torch.log), ("tanh", torch.tanh),
                                         ("sigmoid", torch.sigmoid)]):
        bad = 0
        total = 0
        for rep in range(24):
            if rep == 0:
                x = np.concatenate([specials, np.arange(-120, 120, 1e-3, dtype=np.float32)])
            elif rep < 16:
                x = rng.integers(0, 2 ** 32, 4_000_000, dtype=np.uint64).astype(np.uint32).view(np.float32)    # all bit patterns
            else:
                x = (rng.standard_normal(4_000_000) * (10.0 ** rng.uniform(-3, 2))).astype(np.float32)
            x = np.ascontiguousarray(x)
            n16 = x.size // 16 * 16 + 5            # a length that exercises ATen's vector body and its tail
            x = x[:n16] if x.size >= n16 else x
            y = np.empty_like(x)
            L.run(which, x, y, x.size)
            t = fn(torch.from_numpy(x)).numpy()
            same = (y.view(np.uint32) == t.view(np.uint32)) | (np.isnan(y) & np.isnan(t))
            bad += int((~same).sum())
            total += x.size
            if (~same).any() and bad < 50:
                i = int(np.argmax(~same))
                print(f"  {tname}: x={x[i]!r} ({x[i:i+1].view(np.uint32)[0]:#x}) ours={y[i]!r} torch={t[i]!r}")
        print(f"{tname}: {total - bad} of {total} inputs bit-identical to torch.{tname}")
        ok &= bad == 0
    return ok


if __name__ == "__main__":
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = sys.argv[1:] or [os.path.join(ROOT, "oracle", "c", "pm_sleef_f32.h"),
                            os.path.join(ROOT, "learned-pmctf_amd", "csrc", "pm_sleef_f32.h")]
    for out in outs:                       # the oracle's copy and the product's copy: one text (tests/test_oracle_math.py)
        generate(out)
        print("wrote", out)
    sys.exit(0 if verify(outs[0]) else 1)
