#!/usr/bin/env python3
"""Generates scripts/ubench/vgpr_banks.hip: issue rate of fp32 VALU instructions on gfx950 as a function of where their
operands sit in the VGPR file (register index mod 4).  Explicit registers, eight independent destinations per group."""
import os
VARIANTS = []   # (label, [8 instruction strings])
def bank_regs(bank, n, start=64):     # n registers of the given bank, from v64 upwards (v0..v63 are left to the compiler)
    return [start + 4 * i + bank for i in range(n)]
def add(label, op, dbank, sbanks, acc_first=False, same=None):
    """sbanks: bank of each source; same = (i, j): source j is the SAME register as source i"""
    d = bank_regs(dbank, 8, 64)                       # v64.. destinations
    srcs = [bank_regs(b, 4, 128 + 32 * i) for i, b in enumerate(sbanks)]
    ins = []
    for i in range(8):
        ops = [f"v{srcs[k][(i + k) % 4]}" for k in range(len(sbanks))]
        if same: ops[same[1]] = ops[same[0]]
        if acc_first: ops = [f"v{d[i]}"] + ops
        ins.append(f"{op} v{d[i]}, " + ", ".join(ops))
    VARIANTS.append((label, ins))
add("v_sub   d(0) = a(1) - b(2)", "v_sub_f32", 0, [1, 2])
add("v_sub   d(0) = a(1) - b(1)", "v_sub_f32", 0, [1, 1])
add("v_sub   d(0) = a(0) - b(1)", "v_sub_f32", 0, [0, 1])
add("v_mul   d(0) = a(1) * b(2)", "v_mul_f32", 0, [1, 2])
add("v_fmac  d(0) += a(1) * b(2)", "v_fmac_f32", 0, [1, 2])
add("v_fmac  d(0) += a(0) * b(2)   [src0 = acc]", "v_fmac_f32", 0, [0, 2])
add("v_fmac  d(0) += a(1) * b(0)   [src1 = acc]", "v_fmac_f32", 0, [1, 0])
add("v_fmac  d(0) += a(1) * b(1)   [src0 = src1]", "v_fmac_f32", 0, [1, 1])
add("v_fma   d(3) = a(0) * b(1) + c(2)", "v_fma_f32", 3, [0, 1, 2])
add("v_fma   d(3) = a(0) * b(0) + c(2)   [s0 = s1]", "v_fma_f32", 3, [0, 0, 2])
add("v_fma   d(3) = a(0) * b(1) + c(0)   [s0 = s2]", "v_fma_f32", 3, [0, 1, 0])
add("v_fma   d(3) = a(1) * b(0) + c(0)   [s1 = s2]", "v_fma_f32", 3, [1, 0, 0])
add("v_fma   d(0) = a(0) * b(1) + c(2)   [d = s0]", "v_fma_f32", 0, [0, 1, 2])
add("v_fma   d(0) = a(0) * b(0) + c(0)   [all]", "v_fma_f32", 0, [0, 0, 0])
add("v_fmac  d(0) += a(1) * a(1)   [one register squared]", "v_fmac_f32", 0, [1, 1], same=(0, 1))
add("v_fmac  d(0) += a(0) * a(0)   [squared, bank of acc]", "v_fmac_f32", 0, [0, 0], same=(0, 1))
add("v_fma   d(3) = a(0) * a(0) + c(1)   [squared]", "v_fma_f32", 3, [0, 0, 1], same=(0, 1))
add("v_fma   d(3) = a(0) * a(0) + c(0)   [squared, bank of c]", "v_fma_f32", 3, [0, 0, 0], same=(0, 1))
add("v_fma   d(0) = a(0) * a(0) + c(1)   [squared, bank of d]", "v_fma_f32", 0, [0, 0, 1], same=(0, 1))
add("v_mul   d(0) = a(1) * a(1)   [squared]", "v_mul_f32", 0, [1, 1], same=(0, 1))
add("v_mul   d(0) = a(0) * a(0)   [squared, bank of d]", "v_mul_f32", 0, [0, 0], same=(0, 1))
add("v_sub   d(0) = a(0) - b(0)   [all one bank]", "v_sub_f32", 0, [0, 0])
add("v_min3  d(0) = min3(d, a(1), b(2))", "v_min3_f32", 0, [1, 2], acc_first=True)
add("v_min3  d(0) = min3(d, a(0), b(2))  [s0 = s1]", "v_min3_f32", 0, [0, 2], acc_first=True)
add("v_min3  d(0) = min3(d, a(1), b(1))  [s1 = s2]", "v_min3_f32", 0, [1, 1], acc_first=True)
add("v_min3  d(0) = min3(d, a(0), b(0))  [all]", "v_min3_f32", 0, [0, 0], acc_first=True)
add("v_min   d(0) = min(a(1), b(2))", "v_min_f32", 0, [1, 2])
add("v_min   d(0) = min(a(0), b(0))", "v_min_f32", 0, [0, 0])

def add_pk(label, dlow, alow, blow, clow=None, bcast=False):
    """v_pk_fma_f32 on even-aligned register pairs; *low = index mod 4 of the pair's first register (0 or 2).
    clow None: accumulate in place (c = d).  bcast: src0 uses its low half for both results (op_sel_hi = [0,1,1])."""
    d = [64 + 4 * i + dlow for i in range(8)]
    a = [128 + 4 * i + alow for i in range(4)]
    b = [160 + 4 * i + blow for i in range(4)]
    c = [192 + 4 * i + clow for i in range(4)] if clow is not None else None
    ins = []
    for i in range(8):
        cs = f"v[{c[(i + 2) % 4]}:{c[(i + 2) % 4] + 1}]" if c else f"v[{d[i]}:{d[i] + 1}]"
        mod = " op_sel_hi:[0,1,1]" if bcast else ""
        ins.append(f"v_pk_fma_f32 v[{d[i]}:{d[i] + 1}], v[{a[i % 4]}:{a[i % 4] + 1}], v[{b[(i + 1) % 4]}:{b[(i + 1) % 4] + 1}], {cs}{mod}")
    VARIANTS.append((label, ins))
add_pk("v_pk_fma d(0,1) += a(0,1) * b(2,3)", 0, 0, 2)
add_pk("v_pk_fma d(0,1) += a(2,3) * b(2,3)", 0, 2, 2)
add_pk("v_pk_fma d(2,3) += a(0,1) * b(0,1)", 2, 0, 0)
add_pk("v_pk_fma d(0,1) += a(0,1) * b(0,1)", 0, 0, 0)
add_pk("v_pk_fma d(2,3) += a(0,-) * b(0,1)  [a broadcast]", 2, 0, 0, bcast=True)
add_pk("v_pk_fma d(2,3) += a(2,-) * b(0,1)  [a broadcast]", 2, 2, 0, bcast=True)
add_pk("v_pk_fma d(0,1) += a(2,-) * b(2,3)  [a broadcast]", 0, 2, 2, bcast=True)
add_pk("v_pk_fma d(0,1) = a(0,1) * b(2,3) + c(2,3)", 0, 0, 2, 2)
add_pk("v_pk_fma d(0,1) = a(2,-) * b(0,1) + c(2,3)  [a broadcast]", 0, 2, 0, 2, bcast=True)
add_pk("v_pk_fma d(2,3) = a(2,-) * b(0,1) + c(2,3)  [a broadcast]", 2, 2, 0, 2, bcast=True)
add_pk("v_pk_fma d(2,3) = a(0,-) * b(2,3) + c(0,1)  [a broadcast]", 2, 0, 2, 0, bcast=True)

clob = ",".join(f'"v{i}"' for i in range(64, 256))
init = "\\n".join(f"v_mov_b32 v{i}, %1" for i in range(64, 256))
src = ['// GENERATED by scripts/ubench/gen_vgpr_banks.py -- VGPR bank placement vs issue rate on gfx950', '#include <hip/hip_runtime.h>', '#include <stdio.h>',
       'template <int V> __global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {', '  float x = threadIdx.x * 1e-9f, res = 0.f;', '  const long long t0 = clock64();']
for v, (label, ins) in enumerate(VARIANTS):
    body = "\\n".join(ins)
    src.append(f'  if (V == {v}) asm volatile("{init}\\ns_mov_b32 s20, %2\\n1:\\n.rept 8\\n{body}\\n.endr\\ns_sub_u32 s20, s20, 1\\ns_cmp_lg_u32 s20, 0\\ns_cbranch_scc1 1b\\nv_add_f32 %0, v64, v68\\n" : "=v"(res) : "v"(x), "s"(iters) : {clob}, "s20", "scc");')
src += ['  const long long t1 = clock64();', '  out[blockIdx.x * blockDim.x + threadIdx.x] = res;', '  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;', '}',
        'template <int V> void run(const char* name, int w) {',
        '  const int threads = 256, blocks = 256 * w, iters = 1200000 / w;', '  float* out; (void)hipMalloc(&out, sizeof(float) * threads * blocks);', '  long long* cyc; (void)hipMalloc(&cyc, 8 * 4 * blocks);',
        '  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);',
        '  k<V><<<blocks, threads>>>(out, cyc, iters); (void)hipDeviceSynchronize();',
        '  (void)hipEventRecord(e0); k<V><<<blocks, threads>>>(out, cyc, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);',
        '  float ms; (void)hipEventElapsedTime(&ms, e0, e1);',
        '  const double winstr = (double)blocks * 4 * iters * 64.0;',
        '  printf("%-52s waves/SIMD=%d %8.2f ms  %.2f clk per instr at 2.4 GHz\\n", name, w, ms, 2.4 / (winstr / 1024.0 / (ms * 1e6)));', '  (void)hipFree(cyc);',
        '  (void)hipFree(out);', '}', 'int main() {', '  for (int w : {4}) {']
for v, (label, _) in enumerate(VARIANTS):
    src.append(f'    run<{v}>("{label}", w);')
src += ['  }', '  return 0;', '}']
open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "vgpr_banks.hip"), "w").write("\n".join(src) + "\n")
