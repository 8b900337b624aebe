#!/usr/bin/env python
"""Phase ablation of gemm1x1_kernel: builds three variants of the library under tools/ab/ (git-ignored .so files that
travel with gpurun), each with one phase of the kernel removed, for `DEEPHAR_HIP_LIB=tools/ab/lib_<v>.so python
tools/bench_gemm_shapes.py`:

  noepi   no epilogue (no residual prefetch, no LDS staging, no stores) -- only meaningful for the one-tile-per-wave
          tilings (cfg 13, 16, 17): with more tiles the compiler drops the accumulators that nothing reads
  nomfma  the K loop keeps its DMAs, LDS fragment reads, waits and barriers, the MFMAs become two FMAs
  nodma   the first tile is fetched, every K-step multiplies that same stage: MFMA + LDS + epilogue, no global traffic

Results of round 3: profiles/r03_midsize_gemm_ablation.md."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, 'deephar_amd/csrc/gemm1x1.hip')).read()


def swap(s, old, new):
    assert old in s, old
    return s.replace(old, new)


EPI = "  conv_epilogue<WM, WN, TM, TN, UP2, true>(p, acc, smem, m0, n0, M, epi_vec, pre);"
PRE = "    if (kt == nk - 1) pre.template issue<WM, WN>(p, m0, n0, M, epi_vec);   // lands during the last MFMA block"
MFMA = '''        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);'''
variants = {
    'noepi': swap(swap(src, EPI, "  if (acc[0][0][0] == 123.456f && acc[TM - 1][TN - 1][5] == 1.5f) p.y[tid] = acc[0][0][1];"),
                  PRE, ""),
    'nomfma': swap(src, MFMA, "        acc[i][j][0] += a[i].x * b[j].x; acc[i][j][1] += a[i].w * b[j].w;"),
    'nodma': swap(swap(src, "    if (kt + 1 < nk) issue(kt + 1, cur ^ 1);", "    // (ablation: no DMA after the first tile)"),
                  "    const unsigned so = (unsigned)(cur * STAGE * 4);", "    const unsigned so = 0u;"),
}
os.makedirs(os.path.join(ROOT, 'tools/ab'), exist_ok=True)
for name, text in variants.items():
    os.makedirs('/tmp/ab_src', exist_ok=True)
    path = '/tmp/ab_src/gemm1x1_%s.hip' % name   # a directory of its own: a stray header next to the source would shadow -I
    open(path, 'w').write(text)
    subprocess.run([sys.executable, os.path.join(ROOT, 'tools/build_variant.py'), path,
                    os.path.join(ROOT, 'tools/ab/lib_%s.so' % name)], check=True)
