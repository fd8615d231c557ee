// pdl_bbh.hip — K-bbh: the bidirectional-best-hit filter that the reference's Java host applies to every genome's Scores
// block (ig/infoasys/cli/pangenes/Pangenes.java:98-176), run over the cells where they already are — in HBM, right behind
// K-order — so that only the network edges cross PCIe instead of every cell and both maxima tables.
//
//   phase 1 (:98-128)   a cell (row, col) of two different genomes is a best hit in both directions when its score equals
//                       the row's best against the column's genome (max_genome_score) and the column's best against this
//                       genome (max_genome_score_col); both directions become edges.  inter_max_score[g2] = the largest
//                       such score below 1 per other genome.
//   threshold (:146-155) scoresRowThreshold[row] = min over the row's best-hit edges of inter_max_score[g2]; +inf without one
//   phase 2 (:164-175)  a cell inside one genome (row < col) is kept when it is the best of both genes inside the genome
//                       and not below the row's threshold.
// All comparisons are on the float32 values as computed (bit-identical to the reference's, see pdl_join.hip).
// Edges leave in the host's insertion order: per genome task, phase-1 edges cell by cell ((row, col) then (col, row)),
// then the phase-2 edges.
#include "pdl_common.h"
#include "pdl_scan.h"

struct BbhArgs {
    const float *score; const int32_t *row, *col;
    const uint32_t *taskpos_of, *task_lg, *genome_of;
    const float *MS, *CM;
    uint32_t N, G, Z;
    uint32_t *inter_max;       // [shard][G] float bits, zero-initialised
    uint32_t *thr;             // [rows] float bits, +inf-initialised
    uint8_t *kind;             // [Z] 0: no edge | 1: best hit in both directions | 2: kept intra-genome cell
};

__global__ __launch_bounds__(256) void k_bbh_mark(BbhArgs a) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.Z) return;
    const uint32_t r = (uint32_t) a.row[i], c = (uint32_t) a.col[i];
    const uint32_t p = a.taskpos_of[r], lg = a.task_lg[p];
    const uint32_t g1 = a.genome_of[r], g2 = a.genome_of[c];
    const float s = a.score[i];
    const bool bbh = g1 != g2 && s == a.MS[(size_t) p * a.G + g2] && s == a.CM[(size_t) lg * a.N + c];
    a.kind[i] = bbh ? 1 : 0;
    if (bbh && s < 1.0f) atomicMax(&a.inter_max[(size_t) lg * a.G + g2], __float_as_uint(s));       // (positive floats order like their bits)
}
__global__ __launch_bounds__(256) void k_bbh_threshold(BbhArgs a) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.Z || a.kind[i] != 1) return;
    const uint32_t r = (uint32_t) a.row[i];
    const uint32_t p = a.taskpos_of[r], lg = a.task_lg[p];
    atomicMin(&a.thr[p], a.inter_max[(size_t) lg * a.G + a.genome_of[(uint32_t) a.col[i]]]);
}
__global__ __launch_bounds__(256) void k_bbh_intra(BbhArgs a) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.Z || a.kind[i] == 1) return;
    const uint32_t r = (uint32_t) a.row[i], c = (uint32_t) a.col[i];
    const uint32_t g = a.genome_of[r];
    if (g != a.genome_of[c] || r >= c) return;
    const uint32_t p = a.taskpos_of[r], pc = a.taskpos_of[c];
    const float s = a.score[i];
    if (s == a.MS[(size_t) p * a.G + g] && s == a.MS[(size_t) pc * a.G + g] && s >= __uint_as_float(a.thr[p])) a.kind[i] = 2;
}

struct KindFlag {
    const uint8_t *kind; uint8_t want;
    __device__ uint32_t operator()(uint64_t i) const { return (uint32_t) (kind[i] == want); }
};
struct EdgeApply {          // phase 1 writes two edges per cell, phase 2 one; prefix[i] = cells of the kind before cell i
    const float *score; const int32_t *row, *col;
    int32_t *src, *dst; float *sc; uint32_t *prefix; uint32_t per_cell;
    __device__ void operator()(uint64_t i, uint32_t f, uint32_t pre) const {
        prefix[i] = pre;
        if (!f) return;
        const uint32_t e = pre * per_cell;
        src[e] = row[i]; dst[e] = col[i]; sc[e] = score[i];
        if (per_cell == 2) { src[e + 1] = col[i]; dst[e + 1] = row[i]; sc[e + 1] = score[i]; }
    }
};
__global__ void k_fill_u32(uint32_t *p, uint32_t n, uint32_t v) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v; }
// out[i] = pre1[at[i]], out[n + i] = pre2[at[i]]   (cells of each kind before every genome block)
__global__ void k_pick_prefixes(const uint32_t *pre1, const uint32_t *pre2, const uint32_t *at, uint32_t n, uint32_t z, uint32_t *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && at[i] < z) { out[i] = pre1[at[i]]; out[n + i] = pre2[at[i]]; }
}

// Runs the filter for every genome task of the context; leaves the edges on the host (pinned), per genome
// [phase-1 edges | phase-2 edges].
void pdl_run_bbh_all(pdl_ctx *c) {
    hipStream_t st = c->stream;
    const uint32_t S = (uint32_t) c->shard.size(), n_rows = c->n_task_rows, G = c->G;
    const uint64_t Z = c->h_cell_off.empty() ? 0 : c->h_cell_off.back();
    c->h_edge_off.assign((size_t) S + 1, 0);
    c->h_edge1.assign((size_t) S + 1, 0);
    c->edges_valid = true;
    if (Z == 0 || n_rows == 0) { c->n_edges = 0; return; }
    if (Z >= 0x7fffffffull) PDL_FAIL(PDL_ERR_UNSUPPORTED, "more than 2^31 cells on one device");
    c->bbh_kind.alloc(Z + 16);
    c->bbh_tab.alloc(((size_t) S * G + n_rows + 2 * (Z + 1) + 3 * ((size_t) S + 1)) * sizeof(uint32_t));
    uint32_t *inter_max = c->bbh_tab.as<uint32_t>(), *thr = inter_max + (size_t) S * G, *pre1 = thr + n_rows, *pre2 = pre1 + (Z + 1);
    uint32_t *d_at = pre2 + (Z + 1), *d_pick = d_at + (S + 1);
    PDL_HIP(hipMemsetAsync(inter_max, 0, (size_t) S * G * sizeof(uint32_t), st));
    hipLaunchKernelGGL(k_fill_u32, dim3((n_rows + 255) / 256), dim3(256), 0, st, thr, n_rows, 0x7f800000u);
    BbhArgs a{};
    a.score = c->c_score.as<float>(); a.row = c->c_row.as<int32_t>(); a.col = c->c_col.as<int32_t>();
    a.taskpos_of = c->taskpos_of.as<uint32_t>(); a.task_lg = c->task_lg.as<uint32_t>(); a.genome_of = c->d_gen;
    a.MS = c->MS.as<float>(); a.CM = c->CM.as<float>(); a.N = c->N; a.G = G; a.Z = (uint32_t) Z;
    a.inter_max = inter_max; a.thr = thr; a.kind = c->bbh_kind.as<uint8_t>();
    const uint32_t zb = (uint32_t) ((Z + 255) / 256);
    hipLaunchKernelGGL(k_bbh_mark, dim3(zb), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_bbh_threshold, dim3(zb), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_bbh_intra, dim3(zb), dim3(256), 0, st, a);
    PDL_HIP(hipGetLastError());
    // compaction: phase-1 cells (two edges each) and phase-2 cells (one each), both in cell order
    c->e_src.alloc((2 * Z + Z) * sizeof(int32_t)); c->e_dst.alloc((2 * Z + Z) * sizeof(int32_t)); c->e_score.alloc((2 * Z + Z) * sizeof(float));
    int32_t *src1 = c->e_src.as<int32_t>(), *dst1 = c->e_dst.as<int32_t>(); float *sc1 = c->e_score.as<float>();
    int32_t *src2 = src1 + 2 * Z, *dst2 = dst1 + 2 * Z; float *sc2 = sc1 + 2 * Z;
    uint64_t *d_scal = c->scalars.as<uint64_t>();
    scan_and_apply(c, Z, KindFlag{a.kind, 1}, EdgeApply{a.score, a.row, a.col, src1, dst1, sc1, pre1, 2}, d_scal + 13);
    scan_and_apply(c, Z, KindFlag{a.kind, 2}, EdgeApply{a.score, a.row, a.col, src2, dst2, sc2, pre2, 1}, d_scal + 14);
    // cells of each kind before every genome block: prefix at the block's first cell (the totals close the lists)
    uint64_t tot[2];
    std::vector<uint32_t> h1(S + 1), h2(S + 1);
    c->h_bbh_at.resize(S + 1);
    for (uint32_t i = 0; i <= S; i++) c->h_bbh_at[i] = (uint32_t) c->h_cell_off[i];
    PDL_HIP(hipMemcpyAsync(d_at, c->h_bbh_at.data(), ((size_t) S + 1) * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_pick_prefixes, dim3((S + 1 + 255) / 256), dim3(256), 0, st, pre1, pre2, d_at, S + 1, (uint32_t) Z, d_pick);
    {
        PinRead rd(c);
        const uint64_t *pt = rd.add<uint64_t>(d_scal + 13, 2);
        const uint32_t *pp = rd.add<uint32_t>(d_pick, 2 * ((size_t) S + 1));
        rd.sync();
        tot[0] = pt[0]; tot[1] = pt[1];
        for (uint32_t i = 0; i <= S; i++) {
            const bool past = c->h_cell_off[i] >= Z;
            h1[i] = past ? (uint32_t) tot[0] : pp[i];
            h2[i] = past ? (uint32_t) tot[1] : pp[S + 1 + i];
        }
    }
    const uint64_t n1 = 2 * tot[0], n2 = tot[1];
    c->n_edges = n1 + n2;
    const size_t need = (size_t) (n1 + n2) * 12 + 64;
    if (c->edge_mirror_bytes < need) {
        if (c->edge_mirror) (void) hipHostFree(c->edge_mirror);
        c->edge_mirror = nullptr; c->edge_mirror_bytes = 0;
        PDL_HIP(hipHostMalloc((void **) &c->edge_mirror, need + need / 4, hipHostMallocDefault));
        c->edge_mirror_bytes = need + need / 4;
    }
    uint8_t *m = c->edge_mirror;      // layout: src1 | dst1 | sc1 | src2 | dst2 | sc2
    if (n1) {
        PDL_HIP(hipMemcpyAsync(m, src1, n1 * 4, hipMemcpyDeviceToHost, st));
        PDL_HIP(hipMemcpyAsync(m + n1 * 4, dst1, n1 * 4, hipMemcpyDeviceToHost, st));
        PDL_HIP(hipMemcpyAsync(m + n1 * 8, sc1, n1 * 4, hipMemcpyDeviceToHost, st));
    }
    if (n2) {
        PDL_HIP(hipMemcpyAsync(m + n1 * 12, src2, n2 * 4, hipMemcpyDeviceToHost, st));
        PDL_HIP(hipMemcpyAsync(m + n1 * 12 + n2 * 4, dst2, n2 * 4, hipMemcpyDeviceToHost, st));
        PDL_HIP(hipMemcpyAsync(m + n1 * 12 + n2 * 8, sc2, n2 * 4, hipMemcpyDeviceToHost, st));
    }
    PDL_HIP(hipStreamSynchronize(st));
    c->n_edges1 = n1;
    for (uint32_t i = 0; i <= S; i++) { c->h_edge1[i] = 2ull * h1[i]; c->h_edge_off[i] = h2[i]; }      // phase-1 edge offsets, phase-2 edge offsets
}
