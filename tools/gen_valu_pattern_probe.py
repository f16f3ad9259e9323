"""Generator of tools/valu_pattern_probe.hip (round 5): instruction PATTERNS — not single forms — timed in real shader cycles on gfx950.

tools/valu_clock_probe.hip showed that a wave64 VALU stream issues at ~2.2 cycles per instruction per SIMD in the best case and ~4.1-4.2 in
others, and that which of the two a stream gets depends on the instructions AROUND an instruction as much as on the instruction.  This probe
times whole loop bodies: (A) a reference FMA stream with k "special" instructions (DPP, transcendental, v_mad_u32_u24, readlane, LDS, SALU ...)
placed isolated or in runs, to learn how their cost composes; (B) the step loop of raster_fwd_quad_kernel as hipcc emits it today, and
rearrangements of it, to learn what a better schedule / instruction selection is worth before touching the kernel.

    python tools/gen_valu_pattern_probe.py && hipcc --offload-arch=gfx950 -O3 tools/valu_pattern_probe.hip -o tools/valu_pattern_probe
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ACC = [8, 9, 10, 11, 12, 13, 14, 15]


def F(i):
    n = ACC[i % 8]
    return "v_fma_f32 v%d, v%d, v17, v18" % (n, n)


def rep(special, n_special, n_total, run=False):
    """n_total instructions of which n_special are special(i); isolated (evenly spread) or as one run at the start."""
    out, si = [], 0
    if run:
        pos = set(range(n_special))
    else:
        pos = set(int(round(k * n_total / n_special)) for k in range(n_special))
    for i in range(n_total):
        if i in pos:
            out.append(special(si))
            si += 1
        else:
            out.append(F(i))
    return out


SPECIALS = {
    "v_mov_b32_dpp": lambda i: "v_mov_b32_dpp v%d, v17 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" % (24 + i % 4),
    "v_add_f32_dpp row_shr (in place)": lambda i: "v_add_f32_dpp v%d, v%d, v%d row_shr:1 row_mask:0xf bank_mask:0xf" % ((24 + i % 4,) * 3),
    "v_exp_f32": lambda i: "v_exp_f32 v%d, v19" % (24 + i % 4),
    "v_rcp_f32": lambda i: "v_rcp_f32 v%d, v17" % (24 + i % 4),
    "v_mad_u32_u24": lambda i: "v_mad_u32_u24 v%d, v20, s31, v21" % (24 + i % 4),
    "v_mul_u32_u24": lambda i: "v_mul_u32_u24 v%d, v20, v21" % (24 + i % 4),
    "v_mul_lo_u32": lambda i: "v_mul_lo_u32 v%d, v20, v21" % (24 + i % 4),
    "v_readfirstlane_b32": lambda i: "v_readfirstlane_b32 s22, v17",
    "v_readlane_b32": lambda i: "v_readlane_b32 s22, v17, 16",
    "v_permlane32_swap": lambda i: "v_permlane32_swap_b32 v%d, v%d" % (24 + 2 * (i % 2), 25 + 2 * (i % 2)),
    "v_pk_fma_f32": lambda i: "v_pk_fma_f32 v[%d:%d], v[%d:%d], v[32:33], v[34:35]" % ((24 + 2 * (i % 2), 25 + 2 * (i % 2)) * 2),
    "v_min_f32": lambda i: "v_min_f32 v%d, v17, v%d" % ((24 + i % 4,) * 2),
    "v_cmp_ge_f32 -> sgpr pair": lambda i: "v_cmp_ge_f32_e64 s[20:21], v17, v18",
    "v_cmp_ge_f32 -> vcc": lambda i: "v_cmp_ge_f32 vcc, v17, v18",
    "v_cndmask_b32 (sgpr mask)": lambda i: "v_cndmask_b32_e64 v%d, 0, v17, s[20:21]" % (24 + i % 4),
    "v_cndmask_b32 (vcc)": lambda i: "v_cndmask_b32 v%d, v18, v17, vcc" % (24 + i % 4),
    "v_fma_f32 with sgpr": lambda i: "v_fma_f32 v%d, v%d, s30, v18" % ((24 + i % 4,) * 2),
    "ds_read_b128": lambda i: "ds_read_b128 v[36:39], v22 offset:%d" % (16 * (i % 4)),
    "ds_read_b32": lambda i: "ds_read_b32 v36, v22 offset:%d" % (4 * (i % 4)),
    "ds_read_u8": lambda i: "ds_read_u8 v36, v22",
    "ds_swizzle_b32": lambda i: "ds_swizzle_b32 v%d, v17 offset:0x01F0" % (36 + i % 4),
    "ds_write_b32": lambda i: "ds_write_b32 v22, v17 offset:%d" % (4 * (i % 4)),
    "s_and_b64 (SALU)": lambda i: "s_and_b64 s[12:13], s[20:21], exec",
    "s_nop 1": lambda i: "s_nop 1",
    "s_waitcnt lgkmcnt(0)": lambda i: "s_waitcnt lgkmcnt(0)",
    "v_lshl_add_u32": lambda i: "v_lshl_add_u32 v%d, v20, 4, v21" % (24 + i % 4),
    "v_add_u32": lambda i: "v_add_u32 v%d, v20, v21" % (24 + i % 4),
    "v_cvt_f32_u32": lambda i: "v_cvt_f32_u32 v%d, v20" % (24 + i % 4),
}

patterns = []   # (name, [instructions], n_valu_or_counted)
patterns.append(("reference: 16 x v_fma_f32 vN, vN, v17, v18", [F(i) for i in range(16)]))
for name, sp in SPECIALS.items():
    patterns.append(("1 of 8 isolated: %s" % name, rep(sp, 2, 16)))
    patterns.append(("1 of 16 isolated: %s" % name, rep(sp, 1, 16)))
    patterns.append(("4 of 16, isolated: %s" % name, rep(sp, 4, 16)))
    patterns.append(("4 of 16, one run: %s" % name, rep(sp, 4, 16, run=True)))
    patterns.append(("8 of 32, one run: %s" % name, rep(sp, 8, 32, run=True)))

# ---- (B) the forward step loop -----------------------------------------------------------------------------------------------
# registers: v29 u, v30 v, v34 T, v35 thr (2.0: nothing is taken), v38/v24/v25 colour sums, v3 cur, v9 list address, v8 record base, s42 = 80
FWD_NOW = """s_waitcnt lgkmcnt(0)
v_and_b32 v10, 0x7f, v4
v_mad_u32_u24 v0, v10, s42, v48
ds_read_b128 v[12:15], v0
ds_read_b128 v[16:19], v0 offset:16
ds_read_b128 v[4:7], v0 offset:32
ds_read_b64 v[0:1], v0 offset:48
s_andn2_b64 s[2:3], s[2:3], exec
s_waitcnt lgkmcnt(3)
v_sub_f32 v11, v29, v12
v_sub_f32 v12, v30, v13
s_waitcnt lgkmcnt(1)
v_fma_f32 v5, v5, v12, v18
v_fmac_f32 v5, v4, v11
v_fma_f32 v4, v6, v12, v19
v_fma_f32 v4, v12, v4, 1.0
v_fmac_f32 v4, v11, v5
v_rcp_f32 v4, v4
v_mul_f32 v13, v11, v14
v_mul_f32 v5, v12, v16
v_fmac_f32 v13, v15, v12
v_mul_f32 v5, v5, v5
v_fmac_f32 v5, v13, v13
v_fma_f32 v4, -v5, v4, v17
v_exp_f32_e64 v5, v4 clamp
ds_read_u8 v4, v49
v_mul_f32 v6, v34, v5
v_cmp_ge_f32_e64 s[0:1], v5, v35
s_and_b64 s[12:13], s[0:1], exec
s_or_b64 s[2:3], s[2:3], s[12:13]
v_cndmask_b32_e64 v5, 0, v6, s[0:1]
v_fmac_f32 v34, 0xbf7fbe77, v5
v_cmp_ge_f32 vcc, s47, v34
s_cbranch_vccnz 2f
v_fmac_f32 v38, v7, v5
s_waitcnt lgkmcnt(1)
v_fmac_f32 v24, v0, v5
v_fmac_f32 v25, v1, v5
v_cndmask_b32_e64 v3, v3, v10, s[2:3]
v_add_u32 v40, 1, v40""".split("\n")


def sub(body, old, new):
    out = []
    hit = False
    for ln in body:
        if ln == old:
            hit = True
            out.extend(new)
        else:
            out.append(ln)
    assert hit, old
    return out


patterns.append(("FWD step loop as hipcc emits it (raster_fwd_quad_kernel<pinhole>)", FWD_NOW))
no_lds = [ln for ln in FWD_NOW if not ln.startswith("ds_") and not ln.startswith("s_waitcnt")]
no_lds = sub(no_lds, "v_and_b32 v10, 0x7f, v4", ["v_and_b32 v10, 0x7f, v41"])
patterns.append(("FWD step loop, VALU + SALU only (LDS reads and waits removed)", no_lds))
# (1) the list holds 16-bit byte offsets of the records: ds_read_u16 + v_add_u32 instead of ds_read_u8 + v_and + v_mad_u32_u24
v1 = sub(FWD_NOW, "v_and_b32 v10, 0x7f, v4", ["v_and_b32 v10, 0x1fff, v4"])
v1 = sub(v1, "v_mad_u32_u24 v0, v10, s42, v48", ["v_add_u32 v0, v48, v10"])
patterns.append(("FWD step loop, record address = base + 16-bit offset from the list (no v_mad_u32_u24)", v1))
v1b = sub(FWD_NOW, "v_and_b32 v10, 0x7f, v4", [])
v1b = sub(v1b, "v_mad_u32_u24 v0, v10, s42, v48", ["v_add_u32 v0, v48, v42"])
v1b = sub(v1b, "v_cndmask_b32_e64 v3, v3, v10, s[2:3]", ["v_cndmask_b32_e64 v3, v3, v42, s[2:3]"])
patterns.append(("FWD step loop, address = base + offset already in a register (no v_and, no v_mad)", v1b))
# (2) the same + x*x products from two different registers (v_mul v5, v5, v5 / v_fmac v5, v13, v13 read one register twice)
v2 = sub(v1b, "v_mul_f32 v5, v5, v5", ["v_mul_f32 v5, v5, v43"])
v2 = sub(v2, "v_fmac_f32 v5, v13, v13", ["v_fmac_f32 v5, v13, v44"])
patterns.append(("... + squares formed from two different registers (timing only)", v2))
# (3) the same without the two compares' scalar work
v3 = [ln for ln in v1b if not ln.startswith("s_and_b64") and not ln.startswith("s_or_b64") and not ln.startswith("s_andn2")]
patterns.append(("... (no v_and, no v_mad) without the three SALU mask instructions", v3))
# (4) rcp and exp next to each other (as a two-pixel interleave would allow): timing only, dependencies kept loose
v4 = sub(v1b, "v_rcp_f32 v4, v4", ["v_rcp_f32 v4, v4", "v_exp_f32_e64 v45, v46 clamp"])
v4 = sub(v4, "v_exp_f32_e64 v5, v4 clamp", ["v_mov_b32 v5, v45"])
patterns.append(("... (no v_and, no v_mad) with v_rcp and v_exp adjacent (timing only)", v4))
# (5) without the transcendentals at all (what they cost)
v5 = sub(v1b, "v_rcp_f32 v4, v4", ["v_mov_b32 v4, v4"])
v5 = sub(v5, "v_exp_f32_e64 v5, v4 clamp", ["v_mov_b32 v5, v4"])
patterns.append(("... (no v_and, no v_mad) with v_rcp / v_exp replaced by v_mov (what the two cost)", v5))
# (6) two steps interleaved by hand: A's and B's transcendentals adjacent — approximated by doubling the body with offset registers is not
#     expressible without a register renamer; left to the kernel experiment.

HEADER = r'''// GENERATED by tools/gen_valu_pattern_probe.py — do not edit.  Instruction patterns timed in real shader cycles on gfx950 (round 5).
// hipcc --offload-arch=gfx950 -O3 tools/valu_pattern_probe.hip -o tools/valu_pattern_probe && ./tools/valu_pattern_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <map>
#include <vector>
#define ITERS 3000
#define STR2(x) #x
#define STR(x) STR2(x)
struct WaveRec { uint64_t t0, r0, t1, r1; uint32_t hw_id, xcc_id, pad0, pad1; };
#define CLOBBERS "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", \
    "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", \
    "s0", "s1", "s2", "s3", "s10", "s12", "s13", "s20", "s21", "s22", "s28", "s30", "s31", "s42", "s47", "vcc", "scc", "memory"
#define PROLOGUE                                                                                                                \
    "v_mov_b32 v0, 0\nv_mov_b32 v1, 0\nv_mov_b32 v3, 0\nv_mov_b32 v4, 5\nv_mov_b32 v5, 0\nv_mov_b32 v6, 0\nv_mov_b32 v7, 0\n"      \
    "v_mov_b32 v8, 1.0\nv_mov_b32 v9, 1.0\nv_mov_b32 v10, 1.0\nv_mov_b32 v11, 1.0\nv_mov_b32 v12, 1.0\nv_mov_b32 v13, 1.0\n"         \
    "v_mov_b32 v14, 1.0\nv_mov_b32 v15, 1.0\nv_mov_b32 v16, 1.0\nv_mov_b32 v17, 1.0\nv_mov_b32 v18, 0\nv_mov_b32 v19, 0\n"           \
    "v_mov_b32 v20, 3\nv_mov_b32 v21, 16\nv_mov_b32 v22, 64\nv_mov_b32 v23, 0\nv_mov_b32 v24, 1.0\nv_mov_b32 v25, 1.0\n"             \
    "v_mov_b32 v26, 1.0\nv_mov_b32 v27, 1.0\nv_mov_b32 v28, 0\nv_mov_b32 v29, 0.5\nv_mov_b32 v30, 0.5\nv_mov_b32 v31, 1.0\n"         \
    "v_mov_b32 v32, 1.0\nv_mov_b32 v33, 1.0\nv_mov_b32 v34, 1.0\nv_mov_b32 v35, 2.0\nv_mov_b32 v36, 0\nv_mov_b32 v37, 0\n"           \
    "v_mov_b32 v38, 0\nv_mov_b32 v39, 0\nv_mov_b32 v40, 0\nv_mov_b32 v41, 5\nv_mov_b32 v42, 400\nv_mov_b32 v43, 1.0\n"               \
    "v_mov_b32 v44, 1.0\nv_mov_b32 v45, 0\nv_mov_b32 v46, 0\nv_mov_b32 v47, 0\n"                                                  \
    "v_mov_b32 v48, 16\nv_mov_b32 v49, 0\n"  /* FWD patterns: record base / list address (LDS byte addresses) */                   \
    "s_mov_b32 s30, 1.0\ns_mov_b32 s31, 5\ns_mov_b32 s42, 80\ns_mov_b32 s47, 0x38d1b717\ns_mov_b64 s[20:21], exec\ns_mov_b64 s[2:3], 0\n" \
    "s_mov_b64 vcc, exec\ns_mov_b32 s28, " STR(ITERS) "\n"                                                                          \
    "s_barrier\ns_memtime %0\ns_memrealtime %1\ns_waitcnt lgkmcnt(0)\n1:\n"
#define EPILOGUE                                                                                                                \
    "2:\ns_sub_u32 s28, s28, 1\ns_cmp_lg_u32 s28, 0\ns_cbranch_scc1 1b\ns_waitcnt lgkmcnt(0)\ns_memtime %2\ns_memrealtime %3\ns_waitcnt lgkmcnt(0)\n"
#define PROBE_KERNEL(NAME, BODY)                                                                                                \
    __global__ __launch_bounds__(256) void NAME(WaveRec* out) {                                                                 \
        uint64_t t0, r0, t1, r1;                                                                                                \
        asm volatile(PROLOGUE BODY EPILOGUE : "=&s"(t0), "=&s"(r0), "=&s"(t1), "=&s"(r1) : : CLOBBERS);                         \
        if ((threadIdx.x & 63u) == 0u) {                                                                                        \
            uint32_t hw, xcc;                                                                                                   \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\ns_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));  \
            WaveRec w = {t0, r0, t1, r1, hw, xcc, 0u, 0u};                                                                      \
            out[blockIdx.x * 4u + (threadIdx.x >> 6)] = w;                                                                      \
        }                                                                                                                       \
    }
'''

FOOTER = r'''
typedef void (*kern_t)(WaveRec*);
struct Result { double cyc_per_iter, conc, ghz; int simds; };
// W blocks of 4 waves per CU (dynamic LDS limits the residency); per SIMD: (last s_memtime - first) / (W x ITERS) = cycles per loop iteration
// of ONE wave slot when W waves share the SIMD.  conc = sum of the waves' own intervals / the SIMD's interval (how many really ran together).
static Result run(kern_t k, int W, WaveRec* d, std::vector<WaveRec>& h) {
    const int blocks = 256 * W;
    const size_t lds = (size_t)(160 * 1024 / W) - 1024 > 64 * 1024 ? 64 * 1024 : (size_t)(160 * 1024 / W) - 1024;
    (void)hipMemset(d, 0, sizeof(WaveRec) * blocks * 4);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, d);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, d);
    (void)hipDeviceSynchronize();
    h.resize((size_t)blocks * 4);
    (void)hipMemcpy(h.data(), d, sizeof(WaveRec) * h.size(), hipMemcpyDeviceToHost);
    struct Simd { uint64_t t0 = ~0ull, t1 = 0, sum = 0; int n = 0; };
    std::map<uint32_t, Simd> per_simd;
    std::vector<double> ghz, scyc, conc;
    for (const WaveRec& w : h) {
        const uint32_t simd = (w.hw_id >> 4) & 3u, cu = (w.hw_id >> 8) & 15u, sh = (w.hw_id >> 12) & 1u, se = (w.hw_id >> 13) & 7u;
        Simd& sd = per_simd[(w.xcc_id & 15u) << 16 | se << 12 | sh << 8 | cu << 4 | simd];
        sd.t0 = std::min(sd.t0, w.t0); sd.t1 = std::max(sd.t1, w.t1); sd.n++; sd.sum += w.t1 - w.t0;
        ghz.push_back((double)(w.t1 - w.t0) / ((double)(w.r1 - w.r0) * 10.0));
    }
    for (auto& kv : per_simd)
        if (kv.second.n == W) {
            scyc.push_back((double)(kv.second.t1 - kv.second.t0) / ((double)W * ITERS));
            conc.push_back((double)kv.second.sum / (double)(kv.second.t1 - kv.second.t0));
        }
    std::sort(ghz.begin(), ghz.end()); std::sort(scyc.begin(), scyc.end()); std::sort(conc.begin(), conc.end());
    Result r;
    r.simds = (int)scyc.size();
    r.cyc_per_iter = scyc.empty() ? 0.0 : scyc[scyc.size() / 2];
    r.conc = conc.empty() ? 0.0 : conc[conc.size() / 2];
    r.ghz = ghz[ghz.size() / 2];
    return r;
}
int main() {
    WaveRec* d;
    (void)hipMalloc(&d, sizeof(WaveRec) * 256 * 8 * 4);
    std::vector<WaveRec> h;
    printf("# valu_pattern_probe (MI355X, gfx950): loop bodies timed with s_memtime (shader cycles); W = waves per SIMD; per SIMD and per loop iteration of one wave\n");
    printf("# `instr` = instructions in the body (all kinds); cycles / instr = cycles per iteration / instr.  reference body: v_fma_f32 vN, vN, v17, v18 on 8 accumulators\n");
    printf("| pattern | instr | cycles per iteration W=2 | W=4 | W=8 | cycles / instr W=2 | W=4 | W=8 | waves really concurrent W=8 | clock GHz W=8 |\n|---|---|---|---|---|---|---|---|---|---|\n");
    for (auto& e : ks) {
        Result r2 = run(e.k, 2, d, h), r4 = run(e.k, 4, d, h), r8 = run(e.k, 8, d, h);
        printf("| %s | %d | %.1f | %.1f | %.1f | %.2f | %.2f | %.2f | %.1f | %.3f |\n", e.name, e.n, r2.cyc_per_iter, r4.cyc_per_iter, r8.cyc_per_iter,
               r2.cyc_per_iter / e.n, r4.cyc_per_iter / e.n, r8.cyc_per_iter / e.n, r8.conc, r8.ghz);
        fflush(stdout);
    }
    return 0;
}
'''


def main():
    src = [HEADER]
    table = []
    for i, (name, body) in enumerate(patterns):
        n = sum(1 for ln in body if not ln.endswith(":"))
        src.append("PROBE_KERNEL(k%d, %s)\n" % (i, " ".join('"%s\\n"' % ln for ln in body)))
        table.append('    {"%s", k%d, %d},' % (name.replace('"', "'"), i, n))
    src.append("typedef void (*kern_t)(WaveRec*);\nstatic struct { const char* name; kern_t k; int n; } ks[] = {\n" + "\n".join(table) + "\n};\n")
    src.append(FOOTER.replace("typedef void (*kern_t)(WaveRec*);\n", ""))
    out = os.path.join(ROOT, "tools", "valu_pattern_probe.hip")
    open(out, "w").write("".join(src))
    print("wrote", out, len(patterns), "patterns")


if __name__ == "__main__":
    main()
