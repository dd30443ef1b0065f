// Micro-benchmark: can one SIMD of gfx950 run its matrix pipe and its vector ALU at the same time?  A "unit" is the work of the
// hybrid Chamfer filter for one (query register, 32-reference sub-tile): 4 x v_mfma_f32_32x32x2_f32 (two 32x32 tiles of
// e = |r|^2 - 2 q.r, K = 4) + the in-lane minimum over the 32 accumulator registers (16 v_min3) + the half-wave exchange,
// interleaved with F conflict-free v_fmac_f32 (the pure-VALU formulation of other sub-tiles).  Reported: wall time per unit
// in clk at 2.4 GHz for F = 0 ... 192; perfect overlap = max(MFMA, VALU), none = their sum.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CLOB "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103"
#define FM4(i) "v_fmac_f32 v[130+16*" #i "], v[161+16*" #i "], v[192+16*" #i "]\n v_fmac_f32 v[134+16*" #i "], v[165+16*" #i "], v[196+16*" #i "]\n" \
               "v_fmac_f32 v[138+16*" #i "], v[169+16*" #i "], v[200+16*" #i "]\n v_fmac_f32 v[142+16*" #i "], v[173+16*" #i "], v[204+16*" #i "]\n"
#define FM16 FM4(0) FM4(1) FM4(2) FM4(3)

template <int F, bool MFMA>   // F = fmacs per unit (multiple of 64)
__global__ __launch_bounds__(256) void k(float* out, int units) {
  float x = threadIdx.x * 1e-9f, res;
  asm volatile(
      ".irp n,64,65,66,67,68,69,70,71,72,73,74,75,76,77,78,79,80,81,82,83,84,85,86,87,88,89,90,91,92,93,94,95,96,97,98,99,100,101,102,103\n v_mov_b32 v\\n, %1\n.endr\n"
      ".irp n,128,129,130,131,132,133,134,135,136,137,138,139,140,141,142,143,144,145,146,147,148,149,150,151,152,153,154,155,156,157,158,159,160,161,162,163,164,165,166,167,168,169,170,171,172,173,174,175,176,177,178,179,180,181,182,183,184,185,186,187,188,189,190,191,192,193,194,195,196,197,198,199,200,201,202,203,204,205,206,207,208,209,210,211,212,213,214,215,216,217,218,219,220,221,222,223,224,225,226,227,228,229,230,231,232,233,234,235,236,237,238,239,240,241,242,243,244,245,246,247,248,249,250,251,252,253,254,255\n v_mov_b32 v\\n, %1\n.endr\n"
      "s_mov_b32 s20, %2\n 1:\n"
      ".if %c3\n v_mfma_f32_32x32x2_f32 v[64:79], v96, v98, 0\n .endif\n"
      ".rept %c4\n" FM16 ".endr\n"
      ".if %c3\n v_mfma_f32_32x32x2_f32 v[80:95], v96, v99, 0\n .endif\n"
      ".rept %c4\n" FM16 ".endr\n"
      ".if %c3\n v_mfma_f32_32x32x2_f32 v[64:79], v97, v100, v[64:79]\n .endif\n"
      ".rept %c4\n" FM16 ".endr\n"
      ".if %c3\n v_mfma_f32_32x32x2_f32 v[80:95], v97, v101, v[80:95]\n .endif\n"
      ".rept %c4\n" FM16 ".endr\n"
      ".if %c3\n s_nop 7\n s_nop 7\n s_nop 3\n"
      ".irp n,64,66,68,70,72,74,76,78\n v_min3_f32 v102, v102, v\\n, v[\\n+1]\n.endr\n"
      ".irp n,80,82,84,86,88,90,92,94\n v_min3_f32 v103, v103, v\\n, v[\\n+1]\n.endr\n"
      "v_permlane32_swap_b32 v102, v103\n v_min_f32 v102, v102, v103\n .endif\n"
      "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n v_add_f32 %0, v102, v130\n"
      : "=v"(res) : "v"(x), "s"(units), "n"(MFMA ? 1 : 0), "n"(F / 64)
      : CLOB, "v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139","v140","v141","v142","v143","v144","v145","v146","v147","v148","v149","v150","v151","v152","v153","v154","v155","v156","v157","v158","v159","v160","v161","v162","v163","v164","v165","v166","v167","v168","v169","v170","v171","v172","v173","v174","v175","v176","v177","v178","v179","v180","v181","v182","v183","v184","v185","v186","v187","v188","v189","v190","v191","v192","v193","v194","v195","v196","v197","v198","v199","v200","v201","v202","v203","v204","v205","v206","v207","v208","v209","v210","v211","v212","v213","v214","v215","v216","v217","v218","v219","v220","v221","v222","v223","v224","v225","v226","v227","v228","v229","v230","v231","v232","v233","v234","v235","v236","v237","v238","v239","v240","v241","v242","v243","v244","v245","v246","v247","v248","v249","v250","v251","v252","v253","v254","v255", "s20", "scc");
  out[blockIdx.x * blockDim.x + threadIdx.x] = res;
}

template <int F, bool MFMA>
void run(int w) {
  const int blocks = 256 * w, units = 300000 / w / (1 + F / 64);
  float* out;
  (void)hipMalloc(&out, sizeof(float) * 256 * blocks);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<F, MFMA><<<blocks, 256>>>(out, units); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); k<F, MFMA><<<blocks, 256>>>(out, units); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double clk_per_unit = ms * 1e-3 * 2.4e9 / ((double)units * w);   // per SIMD: w waves share it
  printf("F=%3d fmac/unit  MFMA part %s  waves/SIMD=%d  %8.2f ms  %7.1f clk per unit per SIMD at 2.4 GHz\n", F, MFMA ? "on " : "off", w, ms, clk_per_unit);
  (void)hipFree(out);
}

int main() {
  for (int w : {2, 4}) {
    run<0, true>(w); run<64, false>(w); run<64, true>(w); run<128, false>(w); run<128, true>(w); run<192, false>(w); run<192, true>(w);
    run<256, false>(w); run<256, true>(w);
  }
  return 0;
}
