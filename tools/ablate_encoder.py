"""Build ablated variants of the f16x3 encoder (timing experiments only; results are wrong by construction).

    python tools/ablate_encoder.py            # writes build/abl/lib_<VARIANT>.so
    SAEV_AMD_LIB=build/abl/lib_NOEPI.so python tools/time_encoder.py

Variants: MFMA16, NOIDX, NOSTORE (see below), NOEPI (no TopK epilogue), NOSTAGE (no in-loop operand staging), NOLDS (fragments from registers), NOMFMA (no matrix
instructions), NOBAR (no k-loop barrier), NOCAND (no candidate count/reserve/store), and combinations.  The patches are textual and applied to a temporary copy of the kernel source.
"""
import pathlib
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
SRC = ROOT / "saev_amd" / "csrc" / "gemm_encode_f16x3.hip"
OUT = ROOT / "build" / "abl"
HIPCC = "/opt/rocm/bin/hipcc"


def patched(text: str) -> str:
    def sub(old, new):
        nonlocal text
        assert old in text, old[:60]
        text = text.replace(old, new, 1)

    sub("            if constexpr (WAIT == 8) stage_kstep((t + 3) & 3, s0, kmap(t + 3));",
        "#ifndef ABL_NOSTAGE\n            if constexpr (WAIT == 8) stage_kstep((t + 3) & 3, s0, kmap(t + 3));\n#endif")
    sub("        const float unscale = a.scale_dev != nullptr",
        "        bool skip_epi = false;\n        const float unscale = a.scale_dev != nullptr")
    # NOLDS: fragments come from registers instead of LDS
    sub("            auto load_a = [&](int set, int sb) {",
        "            half8 fconst; for (int e = 0; e < 8; ++e) fconst[e] = (_Float16)(0.01f * (float)((lane + e + t) & 7));\n"
        "            auto load_a = [&](int set, int sb) {\n#ifdef ABL_NOLDS\n                fa[set][0] = fconst; fa[set][1] = fconst; return;\n#endif")
    sub("#pragma unroll\n            for (int jb = 0; jb < 2; ++jb) {\n                fb[jb][0] = *reinterpret_cast<const half8*>(&cs.b[brow0 + 32 * jb][8 * ((0 + half) ^ bsw)]);",
        "#ifdef ABL_NOLDS\n            fb[0][0] = fb[0][1] = fb[1][0] = fb[1][1] = fconst;\n#endif\n#ifndef ABL_NOLDS\n#pragma unroll\n            for (int jb = 0; jb < 2; ++jb) {\n                fb[jb][0] = *reinterpret_cast<const half8*>(&cs.b[brow0 + 32 * jb][8 * ((0 + half) ^ bsw)]);")
    sub("                fb[jb][1] = *reinterpret_cast<const half8*>(&cs.b[brow0 + 32 * jb][8 * ((2 + half) ^ bsw)]);\n            }",
        "                fb[jb][1] = *reinterpret_cast<const half8*>(&cs.b[brow0 + 32 * jb][8 * ((2 + half) ^ bsw)]);\n            }\n#endif")
    sub("        } else {\n            if constexpr (NG == 64) {",
        "        } else {\n#ifdef ABL_NOEPI\n            { float chk = 0.f;\n"
        "              for (int sb = 0; sb < 4; ++sb) for (int jb = 0; jb < 2; ++jb) for (int r = 0; r < 16; ++r) chk += acc[sb][jb][r];\n"
        "              if (chk == 12345.f) a.cand_cnt[0] = 1; }\n            __syncthreads();\n            skip_epi = true;\n#endif\n"
        "            if (!skip_epi) {\n            if constexpr (NG == 64) {")
    # NOBAR: no workgroup barrier inside the k-loop (races: timing only); NOMFMA: no matrix instructions (AR != 0)
    sub("            else asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n            __builtin_amdgcn_s_barrier();",
        "            else asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n#ifndef ABL_NOBAR\n            __builtin_amdgcn_s_barrier();\n#endif")
    sub("    if constexpr (AR == 1) return mfma_bf16(a, b, c);",
        "#ifdef ABL_NOMFMA\n    asm volatile(\"\" :: \"v\"(a), \"v\"(b));\n    return c;\n#endif\n    if constexpr (AR == 1) return mfma_bf16(a, b, c);")
    # MFMA16: every 32x32x16 instruction of the single-product modes becomes two 16x16x32 ones on quarter accumulators (same
    # flops, same fragment reads, wrong results): does the shape's lower power draw (tools/ubench/mfma_issue.hip) survive in
    # the full kernel?
    sub("template <int AR>\n__device__ __forceinline__ f32x16 mfma1(half8 a, half8 b, f32x16 c) {",
        "typedef float f32x4 __attribute__((ext_vector_type(4)));\n"
        "template <int Q>\n__device__ __forceinline__ f32x16 mfma16x2(half8 a, half8 b, f32x16 c) {\n"
        "    f32x4 q0 = {c[8 * Q + 0], c[8 * Q + 1], c[8 * Q + 2], c[8 * Q + 3]}, q1 = {c[8 * Q + 4], c[8 * Q + 5], c[8 * Q + 6], c[8 * Q + 7]};\n"
        "    q0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, q0, 0, 0, 0);\n    q1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, q1, 0, 0, 0);\n"
        "    for (int e = 0; e < 4; ++e) { c[8 * Q + e] = q0[e]; c[8 * Q + 4 + e] = q1[e]; }\n    return c;\n}\n"
        "template <int AR>\n__device__ __forceinline__ f32x16 mfma1(half8 a, half8 b, f32x16 c) {")
    sub("                    acc[sb][0] = mfma1<AR>(fa[as][0], fb[0][0], acc[sb][0]);",
        "#ifdef ABL_MFMA16\n#define MF(Q, A, B, C) mfma16x2<Q>(A, B, C)\n#else\n#define MF(Q, A, B, C) mfma1<AR>(A, B, C)\n#endif\n"
        "                    acc[sb][0] = MF(0, fa[as][0], fb[0][0], acc[sb][0]);")
    sub("                    acc[sb][1] = mfma1<AR>(fa[as][0], fb[1][0], acc[sb][1]);", "                    acc[sb][1] = MF(0, fa[as][0], fb[1][0], acc[sb][1]);")
    sub("                    acc[sb][0] = mfma1<AR>(fa[as][1], fb[0][1], acc[sb][0]);", "                    acc[sb][0] = MF(1, fa[as][1], fb[0][1], acc[sb][0]);")
    sub("                    acc[sb][1] = mfma1<AR>(fa[as][1], fb[1][1], acc[sb][1]);", "                    acc[sb][1] = MF(1, fa[as][1], fb[1][1], acc[sb][1]);")
    # NOIDX: candidate values are stored, their indices are not (one store and the index arithmetic less per candidate);
    # NOSTORE: neither (compare / exec mask / offset bookkeeping stay)
    sub("                                *reinterpret_cast<float*>(reinterpret_cast<char*>(a.cand_val) + off) = v;\n",
        "#ifndef ABL_NOSTORE\n                                *reinterpret_cast<float*>(reinterpret_cast<char*>(a.cand_val) + off) = v;\n#endif\n"
        "#if !defined(ABL_NOSTORE) && !defined(ABL_NOIDX)\n")
    sub("                                    s0 + ws * 128 + sb * 32 + 8 * (r >> 2) + 4 * half + (r & 3);\n",
        "                                    s0 + ws * 128 + sb * 32 + 8 * (r >> 2) + 4 * half + (r & 3);\n#endif\n")
    # NOATOM (16x16x32 kernel): list space is not reserved (no global atomic; positions are made up)
    sub("            if (kg == 0 && rowtot[jb] > 0) base[jb] = atomicAdd(&a.cand_cnt[b], rowtot[jb]);",
        "#ifdef ABL_NOATOM\n            base[jb] = tile_no * 24;\n#else\n            if (kg == 0 && rowtot[jb] > 0) base[jb] = atomicAdd(&a.cand_cnt[b],  rowtot[jb]);\n#endif")
    # NOST: the mask walk runs (park, ds_read, index arithmetic) but nothing is stored to the lists
    sub("                *reinterpret_cast<float*>(reinterpret_cast<char*>(a.cand_val) + off) = v;\n"
        "                *reinterpret_cast<int32_t*>(reinterpret_cast<char*>(a.cand_idx) + off) = lat0 + 16 * c + e;\n",
        "#ifdef ABL_NOST\n                asm volatile(\"\" :: \"v\"(v), \"v\"(lat0 + 16 * c + e), \"v\"(off));\n#else\n"
        "                *reinterpret_cast<float*>(reinterpret_cast<char*>(a.cand_val) + off) = v;\n"
        "                *reinterpret_cast<int32_t*>(reinterpret_cast<char*>(a.cand_idx) + off) = lat0 + 16 * c + e;\n#endif\n")
    # SLEEP1 / SLEEP2: ~1 / ~2 us of s_sleep after the candidate stores of every tile -- if the kernel does not get slower by
    # 32 x that, the time was already being spent waiting for the stores' acknowledgements
    sub("        if (!prefetched && st + 1 < st_end) {  // (never",
        "#ifdef ABL_SLEEP1\n        __builtin_amdgcn_s_sleep(30);\n#endif\n#ifdef ABL_SLEEP2\n        __builtin_amdgcn_s_sleep(60);\n#endif\n"
        "        if (!prefetched && st + 1 < st_end) {  // (never")
    # NOIDX16: the mask walk stores the values only (half the list bytes): is the cost of the stores their bytes?
    sub("                *reinterpret_cast<int32_t*>(reinterpret_cast<char*>(a.cand_idx) + off) = lat0 + 16 * c + e;\n#endif\n",
        "#ifndef ABL_NOIDX16\n                *reinterpret_cast<int32_t*>(reinterpret_cast<char*>(a.cand_idx) + off) = lat0 + 16 * c + e;\n#endif\n#endif\n")
    # NOEMIT: hit masks are formed and space is reserved, nothing is parked or stored
    sub("            uint32_t mm = (row_base + rowtot[jb] <= a.cand_cap) ? hit[jb] : 0u;",
        "#ifdef ABL_NOEMIT\n            uint32_t mm = (sm.tau_key[1] == 777777 && row_base + rowtot[jb] <= a.cand_cap) ? hit[jb] : 0u;\n#else\n"
        "            uint32_t mm = (row_base + rowtot[jb] <= a.cand_cap) ? hit[jb] : 0u;\n#endif")
    # NOEPI16: the 16x16x32 kernel without its epilogue (NOST / NOEMIT below: without the list stores / the whole mask walk)
    sub("        // lane owns batch rows bl(jb) = wb*64 + jb*16 + l15; latent of acc[sb][jb][e]: sl = ws*128 + sb*16 + 4*kg + e\n",
        "#ifdef ABL_NOEPI16\n        { float chk = 0.f;\n          for (int sb = 0; sb < 8; ++sb) for (int jb = 0; jb < 4; ++jb) for (int e = 0; e < 4; ++e) chk += acc[sb][jb][e];\n"
        "          if (chk == 12345.f) a.cand_cnt[0] = 1; }\n        if (sm.tau_key[1] == 777777)\n#endif\n        {\n")
    sub("        if (!prefetched && st + 1 < st_end) {  // (never", "        }\n        if (!prefetched && st + 1 < st_end) {  // (never")
    sub("            int npass[2], pos[2];\n",
        "#ifdef ABL_NOCAND\n            if (sm.tau_key[0] == 12345) a.cand_cnt[0] = (int)acc[0][0][0] + (int)acc[1][1][1] + (int)acc[2][0][2] + (int)acc[3][1][3];\n"
        "            if (sm.tau_key[1] != 777777) goto tile_done;\n#endif\n            int npass[2], pos[2];\n")
    sub("        if (!prefetched && st + 1 < st_end) {",
        "            }\n#ifdef ABL_NOCAND\n        tile_done:;\n#endif\n        if (!prefetched && st + 1 < st_end) {")
    # the extra '}' above closes `if (!skip_epi) {`; it must sit inside the else-branch: move the else's own brace
    sub("            }\n        }\n            }\n#ifdef ABL_NOCAND", "            }\n            }\n        }\n#ifdef ABL_NOCAND")
    return text


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    tmp = OUT / "gemm_encode_f16x3_abl.hip"
    tmp.write_text(patched(SRC.read_text()))
    variants = sys.argv[1:] or ["NOEPI", "NOCAND", "NOSTAGE+NOEPI", "NOLDS+NOEPI", "NOSTAGE+NOLDS+NOEPI", "NOMFMA+NOEPI", "NOMFMA+NOLDS+NOEPI", "NOMFMA+NOSTAGE+NOEPI"]
    objs = [str(ROOT / "build" / f"{n}.o") for n in ("ctx", "gemm_encode", "split", "select", "sparse", "tail", "auxk")]
    for v in variants:
        defs = [f"-DABL_{x}" for x in v.split("+")]
        obj = OUT / f"f16_{v}.o"
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed", "-Wno-unused-value",
                               f"-I{ROOT / 'include'}", f"-I{ROOT / 'saev_amd' / 'csrc'}", *defs, "-c", str(tmp), "-o", str(obj)])
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, str(obj),
                               "-o", str(OUT / f"lib_{v}.so")])
        print("built", OUT / f"lib_{v}.so")


if __name__ == "__main__":
    main()
