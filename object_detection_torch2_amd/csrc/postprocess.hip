// Inference post-process: box decode + score (utils.py:19-55), pairwise IoU (utils.py:58-77) and the greedy
// class-agnostic NMS (utils.py:80-116).  Built with -ffp-contract=off so IoU comparisons use the reference's
// separately-rounded fp32 ops.
//
// NMS: one 1024-thread workgroup per image, everything in LDS (160 KiB):
//   1. candidates (max over class cols 5: > 0) compacted as 64-bit keys  score_bits<<32 | ~index
//   2. bitonic sort, descending (ties -> lower index first)
//   3. 64-candidate chunks: all 16 waves test the chunk against the kept list (LDS broadcast reads) and build the
//      64x64 intra-chunk IoU bit-matrix with __ballot; wave 0 resolves the serial greedy chain on scalar bitmasks
//   4. rows not kept get their score columns zeroed in place (utils.py:114)
#include "common.h"

namespace {

__device__ __forceinline__ bool iou_gt(float4 t, float4 s, float thr) {  // utils.py:74-77, strict '>'
    float tw2 = t.z / 2.0f, sw2 = s.z / 2.0f, th2 = t.w / 2.0f, sh2 = s.w / 2.0f;
    float w = fminf(t.x + tw2, s.x + sw2) - fmaxf(t.x - tw2, s.x - sw2);
    w = fmaxf(w, 0.0f);
    float h = fminf(t.y + th2, s.y + sh2) - fmaxf(t.y - th2, s.y - sh2);
    h = fmaxf(h, 0.0f);
    float wh = w * h;
    float iou = wh > 0.0f ? wh / (t.z * t.w + s.z * s.w - wh) : wh;
    return iou > thr;
}

__device__ __forceinline__ float iou_val(const float* t, const float* s) {
    float tw2 = t[2] / 2.0f, sw2 = s[2] / 2.0f, th2 = t[3] / 2.0f, sh2 = s[3] / 2.0f;
    float w = fminf(t[0] + tw2, s[0] + sw2) - fmaxf(t[0] - tw2, s[0] - sw2);
    w = fmaxf(w, 0.0f);
    float h = fminf(t[1] + th2, s[1] + sh2) - fmaxf(t[1] - th2, s[1] - sh2);
    h = fmaxf(h, 0.0f);
    float wh = w * h;
    return wh > 0.0f ? wh / (t[2] * t[3] + s[2] * s[3] - wh) : wh;
}

template <int NC>
__global__ __launch_bounds__(256) void decode_score_kernel(float* __restrict__ outputs, const float* __restrict__ dboxes, int P,
                                                           int64_t total) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int p = (int)(idx % P);
    float* row = outputs + idx * (4 + NC);
    float4 d = reinterpret_cast<const float4*>(dboxes)[p];
    float v[4 + NC];
#pragma unroll
    for (int c = 0; c < 4 + NC; ++c) v[c] = row[c];
    // utils.calc_coordicate, utils.py:35-38 (mul then add, no fma)
    float cx = d.z * v[0] + d.x;
    float cy = d.w * v[1] + d.y;
    float w = d.z * expf(v[2]);
    float h = d.w * expf(v[3]);
    // utils.calc_score, utils.py:54-55: softmax * onehot(argmax) -> one non-zero per row
    float mx = v[4];
    int arg = 0;
#pragma unroll
    for (int c = 1; c < NC; ++c)
        if (v[4 + c] > mx) { mx = v[4 + c]; arg = c; }
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < NC; ++c) s += expf(v[4 + c] - mx);
    float score = expf(v[4 + arg] - mx) / s;
    row[0] = cx; row[1] = cy; row[2] = w; row[3] = h;
#pragma unroll
    for (int c = 0; c < NC; ++c) row[4 + c] = (c == arg) ? score : 0.0f;
}

// any number of classes: the same arithmetic as decode_score_kernel<NC> with the row read twice instead of held in registers
__global__ __launch_bounds__(256) void decode_score_rt_kernel(float* __restrict__ outputs, const float* __restrict__ dboxes, int P, int NC,
                                                              int64_t total) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int p = (int)(idx % P);
    float* row = outputs + idx * (4 + NC);
    float4 d = reinterpret_cast<const float4*>(dboxes)[p];
    float cx = d.z * row[0] + d.x;
    float cy = d.w * row[1] + d.y;
    float w = d.z * expf(row[2]);
    float h = d.w * expf(row[3]);
    float mx = row[4];
    int arg = 0;
    for (int c = 1; c < NC; ++c)
        if (row[4 + c] > mx) { mx = row[4 + c]; arg = c; }
    float s = 0.0f;
    for (int c = 0; c < NC; ++c) s += expf(row[4 + c] - mx);
    float score = expf(row[4 + arg] - mx) / s;
    row[0] = cx; row[1] = cy; row[2] = w; row[3] = h;
    for (int c = 0; c < NC; ++c) row[4 + c] = (c == arg) ? score : 0.0f;
}

__global__ void calc_iou_kernel(const float* __restrict__ t, const float* __restrict__ s, int T, int S, int rowt, int rows,
                                float* __restrict__ iou) {
    const int n = blockIdx.z;
    const int si = blockIdx.x * blockDim.x + threadIdx.x;
    const int ti = blockIdx.y;
    if (si >= S) return;
    const float* tp = t + ((size_t)n * T + ti) * rowt;
    const float* sp = s + ((size_t)n * S + si) * rows;
    iou[((size_t)n * T + ti) * S + si] = iou_val(tp, sp);
}

// True-positive assignment of the evaluation loop (evaluate.py:134-151), one block per image.  In the reference, per class c:
// predictions with a positive class-c score in descending score order, GT rows of class c; every prediction claims its best-IoU
// GT (first maximum) if IoU > thr; of the claimants of a GT only the first in score order is a true positive
// ((valid.cumsum(0) == valid) * valid).  After utils.calc_score every row has exactly one positive score column, so the classes
// partition the predictions and all of them are handled in one pass: the winner of a GT is found by a 64-bit atomic max on
// (score bits, ~row), i.e. highest score, ties to the lower row (the reference's torch.sort leaves ties unspecified).
__global__ __launch_bounds__(1024) void eval_match_kernel(const float* __restrict__ outputs, const float* __restrict__ gts, int P, int G,
                                                          int C, float thr, int* __restrict__ cls, uint8_t* __restrict__ tp,
                                                          float* __restrict__ score, int* __restrict__ gt_count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* win = reinterpret_cast<unsigned long long*>(smem);          // G winners
    float* gbox = reinterpret_cast<float*>(smem + (size_t)G * 8);                    // G x 4
    int* gcls = reinterpret_cast<int*>(smem + (size_t)G * 24);                       // G
    const int n = blockIdx.x, row = 4 + C;
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        const float* t = gts + ((size_t)n * G + g) * row;
        int c = -1;
        for (int k = 0; k < C - 1; ++k)
            if (t[5 + k] > 0.f) { c = k; break; }
        gcls[g] = c;
        gbox[4 * g] = t[0]; gbox[4 * g + 1] = t[1]; gbox[4 * g + 2] = t[2]; gbox[4 * g + 3] = t[3];
        win[g] = 0ull;
        if (c >= 0 && gt_count) atomicAdd(&gt_count[c], 1);
    }
    __syncthreads();
    auto claim = [&](int p, int& c, unsigned long long& key) -> int {      // class / key of prediction p; returns its claimed GT or -1
        const float* o = outputs + ((size_t)n * P + p) * row;
        c = -1;
        float score = 0.f;
        for (int k = 0; k < C - 1; ++k)
            if (o[5 + k] > 0.f) { c = k; score = o[5 + k]; break; }
        if (c < 0) return -1;
        int best = -1;
        float biou = 0.f;
        for (int g = 0; g < G; ++g) {
            if (gcls[g] != c) continue;
            const float v = iou_val(o, gbox + 4 * g);
            if (best < 0 || v > biou) { best = g; biou = v; }
        }
        key = ((unsigned long long)__float_as_uint(score) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)p);
        return (best >= 0 && biou > thr) ? best : -1;
    };
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
        int c;
        unsigned long long key;
        const int g = claim(p, c, key);
        if (g >= 0) atomicMax(&win[g], key);
    }
    __syncthreads();
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
        int c;
        unsigned long long key = 0;
        const int g = claim(p, c, key);
        cls[(size_t)n * P + p] = c;
        tp[(size_t)n * P + p] = (g >= 0 && win[g] == key) ? 1 : 0;
        if (score) score[(size_t)n * P + p] = c >= 0 ? __uint_as_float((unsigned)(key >> 32)) : 0.f;
    }
}

constexpr int NMS_THREADS = 1024;
constexpr int NMS_WAVES = NMS_THREADS / 64;
constexpr int NMS_LDS_BYTES = 160 * 1024;
constexpr int NMS_MISC_BYTES = 64 * 8 + NMS_WAVES * 8 + 64;  // rowmask[64], wmask[16], scalars
constexpr int NMS_WS_SMALL_P = 9984;     // up to here the LDS-resident sort / scan kernels of the workspace variant apply
constexpr int NMS_WS_MAX_P = 32768;      // large-P kernels (SSD512: 24,564 priors): chunked sort, scan on global mask rows
constexpr int SORT_CHUNK = 16384;        // keys per LDS residency of the chunked bitonic sort (128 KiB)

__global__ __launch_bounds__(NMS_THREADS) void nms_kernel(float* __restrict__ outputs, int P, int C, float thr,
                                                          uint8_t* __restrict__ keep_out, int idx_bytes, int kept_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int n = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ROW = 4 + C;
    float* out = outputs + (size_t)n * P * ROW;

    unsigned long long* keys = reinterpret_cast<unsigned long long*>(lds);
    // regions valid after the sort: idx16 | kept boxes | keepbits | misc
    unsigned short* idx16 = reinterpret_cast<unsigned short*>(lds);
    float4* kept = reinterpret_cast<float4*>(lds + idx_bytes);
    unsigned char* tail = lds + NMS_LDS_BYTES - NMS_MISC_BYTES;
    unsigned long long* rowmask = reinterpret_cast<unsigned long long*>(tail);
    unsigned long long* wmask = rowmask + 64;
    int* scal = reinterpret_cast<int*>(wmask + NMS_WAVES);  // [0] candidate count, [1] kept count
    unsigned* keepbits = reinterpret_cast<unsigned*>(tail) - ((P + 31) / 32 + 3) / 4 * 4;

    if (tid == 0) { scal[0] = 0; scal[1] = 0; }
    __syncthreads();
    // 1. candidate keys, compacted (order irrelevant before the sort; keys are unique)
    for (int i = tid; i < P; i += NMS_THREADS) {
        const float* row = out + (size_t)i * ROW;
        float sc = row[5];
        for (int c = 6; c < ROW; ++c) sc = fmaxf(sc, row[c]);   // torch.max(output[:, 5:], dim=1), utils.py:99
        if (sc > 0.0f) {
            int slot = atomicAdd(&scal[0], 1);
            keys[slot] = ((unsigned long long)__float_as_uint(sc) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
        }
    }
    __syncthreads();
    const int K = scal[0];
    int sortn = 64;
    while (sortn < K) sortn <<= 1;
    for (int i = K + tid; i < sortn; i += NMS_THREADS) keys[i] = 0ull;
    __syncthreads();
    // 2. bitonic sort, descending
    for (int k = 2; k <= sortn; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (sortn >> 1); t += NMS_THREADS) {
                int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                int l = i | j;
                unsigned long long a = keys[i], b = keys[l];
                bool desc = (i & k) == 0;
                if ((a < b) == desc) { keys[i] = b; keys[l] = a; }
            }
            __syncthreads();
        }
    }
    // 3. compact to 16-bit prior indices (registers first: idx16 aliases keys)
    unsigned short myidx[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int i = tid + r * NMS_THREADS;
        myidx[r] = (i < K) ? (unsigned short)(0xFFFFFFFFu - (unsigned)(keys[i] & 0xFFFFFFFFull)) : (unsigned short)0;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int i = tid + r * NMS_THREADS;
        if (i < K) idx16[i] = myidx[r];
    }
    for (int i = tid; i < (P + 31) / 32; i += NMS_THREADS) keepbits[i] = 0u;
    __syncthreads();
    // 4. greedy suppression, 64 sorted candidates per chunk
    const int nchunks = (K + 63) >> 6;
    for (int c = 0; c < nchunks; ++c) {
        const int ci = c * 64 + lane;
        const bool valid = ci < K;
        const int pidx = valid ? (int)idx16[ci] : 0;
        float4 box = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) {
            const float* row = out + (size_t)pidx * ROW;
            box = make_float4(row[0], row[1], row[2], row[3]);
        }
        const int nkept = scal[1];
        // (a) against boxes kept in earlier chunks; waves stride over the kept list
        bool sup = false;
        for (int j = wave; j < nkept; j += NMS_WAVES) {
            float4 kb = kept[j];
            sup |= iou_gt(kb, box, thr);
        }
        unsigned long long sm = __ballot(sup && valid);
        if (lane == 0) wmask[wave] = sm;
        // (b1) intra-chunk matrix: wave w owns heads 4w..4w+3; bit j of rowmask[i] = head i suppresses later row j
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = wave * 4 + r;
            float4 hb;
            hb.x = __shfl(box.x, i, 64); hb.y = __shfl(box.y, i, 64); hb.z = __shfl(box.z, i, 64); hb.w = __shfl(box.w, i, 64);
            bool hit = valid && (lane > i) && iou_gt(hb, box, thr);
            unsigned long long m = __ballot(hit);
            if (lane == 0) rowmask[i] = m;
        }
        __syncthreads();
        if (wave == 0) {
            unsigned long long supm = 0ull;
#pragma unroll
            for (int w = 0; w < NMS_WAVES; ++w) supm |= wmask[w];
            unsigned long long alive = __ballot(valid) & ~supm;
            const unsigned long long mine = rowmask[lane];
            const unsigned mlo = (unsigned)mine, mhi = (unsigned)(mine >> 32);
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                unsigned lo = __builtin_amdgcn_readlane(mlo, i);
                unsigned hi = __builtin_amdgcn_readlane(mhi, i);
                unsigned long long rm = ((unsigned long long)hi << 32) | lo;
                if ((alive >> i) & 1ull) alive &= ~rm;
            }
            const bool is_kept = (alive >> lane) & 1ull;
            if (is_kept) {
                int pos = nkept + (int)__popcll(alive & ((1ull << lane) - 1ull));
                if (pos < kept_cap) kept[pos] = box;
                atomicOr(&keepbits[pidx >> 5], 1u << (pidx & 31));
            }
            if (lane == 0) scal[1] = nkept + (int)__popcll(alive);
        }
        __syncthreads();
    }
    // 5. outputs[:, :, 4:] *= keep mask (utils.py:114): suppressed, void-argmax and non-candidate rows end all-zero
    for (int i = tid; i < P; i += NMS_THREADS) {
        bool k = (keepbits[i >> 5] >> (i & 31)) & 1u;
        if (keep_out) keep_out[(size_t)n * P + i] = k ? 1 : 0;
        if (!k) {
            float* row = out + (size_t)i * ROW;
            for (int c = 4; c < ROW; ++c) row[c] = row[c] * 0.0f;
        }
    }
}

// ---- NMS, workspace variant: the O(K^2) IoU work spread over the whole chip ---------------------------------------------
//   nms_sort_kernel   per image: candidate keys -> LDS bitonic sort -> sorted prior indices + boxes + count in the workspace
//   nms_mask_kernel   64-row x 256-column tiles of the upper triangle: lane = column candidate, 64 head rows broadcast by
//                     shuffle, one __ballot per (row, column chunk) -> bit-matrix mask[n][row][chunk]
//   nms_scan_kernel   per image: wave 0 walks the sorted candidates; `removed` bits live in registers (3 words per lane),
//                     rows are prefetched 4 ahead; kept rows are scattered to a flag array, then the block zeroes the
//                     score columns of every prior that is not kept (utils.py:114)
__device__ __forceinline__ int argmax_class(const float* row, int ROW) {      // first maximum of the non-void score columns
    int best = 5;
    float sc = row[5];
    for (int c = 6; c < ROW; ++c)
        if (row[c] > sc) { sc = row[c]; best = c; }
    return best;
}

struct NmsWs {
    int* count;            // (N)
    int* sidx;             // (N, P) sorted prior indices
    float4* sbox;          // (N, P) boxes in sorted order
    unsigned char* flag;   // (N, P) keep flags by prior index
    unsigned long long* mask;   // (N, P, W) W = ceil(P / 64)
    unsigned long long* keys;   // (N, sortcap) sort keys in global memory, large-P variant only (P > NMS_WS_SMALL_P)
    int* scls;             // (N, P) class (arg-max score column) of the sorted candidates, read by the per-class option
    int W, sortcap, per_class;
};

__global__ __launch_bounds__(NMS_THREADS) void nms_sort_kernel(const float* __restrict__ outputs, int P, int C, NmsWs ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(lds);
    __shared__ int cnt;
    const int n = blockIdx.x, tid = threadIdx.x;
    const int ROW = 4 + C;
    const float* out = outputs + (size_t)n * P * ROW;
    if (tid == 0) cnt = 0;
    __syncthreads();
    for (int i = tid; i < P; i += NMS_THREADS) {
        const float* row = out + (size_t)i * ROW;
        float sc = row[5];
        for (int c = 6; c < ROW; ++c) sc = fmaxf(sc, row[c]);
        ws.flag[(size_t)n * P + i] = 0;
        if (sc > 0.0f) {
            int slot = atomicAdd(&cnt, 1);
            keys[slot] = ((unsigned long long)__float_as_uint(sc) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
        }
    }
    __syncthreads();
    const int K = cnt;
    int sortn = 64;
    while (sortn < K) sortn <<= 1;
    for (int i = K + tid; i < sortn; i += NMS_THREADS) keys[i] = 0ull;
    __syncthreads();
    for (int k = 2; k <= sortn; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (sortn >> 1); t += NMS_THREADS) {
                int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                int l = i | j;
                unsigned long long a = keys[i], b = keys[l];
                bool desc = (i & k) == 0;
                if ((a < b) == desc) { keys[i] = b; keys[l] = a; }
            }
            __syncthreads();
        }
    }
    if (tid == 0) ws.count[n] = K;
    for (int i = tid; i < K; i += NMS_THREADS) {
        const int idx = (int)(0xFFFFFFFFu - (unsigned)(keys[i] & 0xFFFFFFFFull));
        const float* row = out + (size_t)idx * ROW;
        ws.sidx[(size_t)n * P + i] = idx;
        ws.sbox[(size_t)n * P + i] = make_float4(row[0], row[1], row[2], row[3]);
        ws.scls[(size_t)n * P + i] = argmax_class(row, ROW);
    }
}

__global__ __launch_bounds__(256) void nms_mask_kernel(int P, float thr, NmsWs ws) {
    const int n = blockIdx.z, rc = blockIdx.y, cg = blockIdx.x;       // row chunk (64 rows), column group (4 chunks)
    const int K = ws.count[n];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int cc = cg * 4 + wave;                                      // this wave's column chunk
    if (rc * 64 >= K || cc < rc || cc * 64 >= K) return;               // upper triangle (incl. diagonal) of live chunks only
    const float4* sb = ws.sbox + (size_t)n * P;
    const int j = cc * 64 + lane, i0 = rc * 64;
    const bool jv = j < K;
    const float4 bj = jv ? sb[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 bi = (i0 + lane < K) ? sb[i0 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int* sc = ws.scls + (size_t)n * P;
    const int cj = (ws.per_class && jv) ? sc[j] : 0, ci = (ws.per_class && i0 + lane < K) ? sc[i0 + lane] : 0;
    unsigned long long mine = 0ull;
#pragma unroll 8
    for (int r = 0; r < 64; ++r) {
        float4 hb;
        hb.x = __shfl(bi.x, r, 64); hb.y = __shfl(bi.y, r, 64); hb.z = __shfl(bi.z, r, 64); hb.w = __shfl(bi.w, r, 64);
        const bool same = !ws.per_class || __shfl(ci, r, 64) == cj;       // per-class option: only same-class boxes suppress each other
        const bool hit = jv && (j > i0 + r) && same && iou_gt(hb, bj, thr);
        const unsigned long long m = __ballot(hit);
        if (lane == r) mine = m;
    }
    if (i0 + lane < K) ws.mask[((size_t)n * P + i0 + lane) * ws.W + cc] = mine;
}

__global__ __launch_bounds__(256) void nms_scan_kernel(float* __restrict__ outputs, int P, int C, NmsWs ws, unsigned char* __restrict__ keep_out) {
    // Greedy chain over the precomputed bit-matrix, 64 candidates (one mask word) per batch:
    //   waves 0-2  lane l of wave w owns suppression word 64w + l ("removed" bits of candidates 64*(64w+l) ..) in a register
    //   wave 3     streams the next batch of 64 mask rows from the workspace into the other LDS buffer (16 B per lane)
    //   per batch  wave 0 resolves the 64x64 diagonal block on scalar bitmasks (v_readlane with constant lane, like the LDS
    //              variant), then every owner lane ORs the rows of the surviving candidates into its word (LDS reads, no
    //              serial dependency); two barriers per batch, global latency never on the chain
    extern __shared__ __attribute__((aligned(16))) unsigned long long rows[];     // [2][64][W] + 2 scalars
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = ws.count[n];
    const int ROW = 4 + C;
    const int W = ws.W, WK = (K + 63) >> 6;       // row stride in words / words the mask kernel wrote for this image
    const unsigned long long* mrow = ws.mask + (size_t)n * P * W;
    unsigned long long* slot = rows + (size_t)2 * 64 * W;      // [0] removed word of the current batch, [1] its survivors
    const int myword = wave * 64 + lane;                       // waves 0-2
    unsigned long long rem = 0ull, kept = 0ull;

    // wave 3: rows [64bt, 64bt+64) straight into LDS with LDS-DMA (global_load_lds, 16 B per lane, no VGPR staging, all
    // 64-128 transfers of a batch in flight at once).  Destination = wave-uniform row base + lane * 16 B, so lanes whose
    // word pair lies outside [bt, WK) are simply masked off (EXEC) and leave their LDS slot untouched (never read).
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    auto load_batch = [&](int bt) {
        unsigned long long* dst = rows + (size_t)(bt & 1) * 64 * W;
        const int nrows = min(64, K - bt * 64);
        const int wp0 = 2 * lane, wp1 = 2 * lane + 128;
        const bool a0 = wp0 + 1 >= bt && wp0 < WK, a1 = wp1 + 1 >= bt && wp1 < WK;
        for (int r = 0; r < nrows; ++r) {
            const unsigned long long* g = mrow + (size_t)(bt * 64 + r) * W;
            if (a0) __builtin_amdgcn_global_load_lds((gptr_t)(g + wp0), (lptr_t)(dst + r * W), 16, 0, 0);
            if (a1) __builtin_amdgcn_global_load_lds((gptr_t)(g + wp1), (lptr_t)(dst + r * W + 128), 16, 0, 0);
        }
    };
    if (wave == 3 && WK > 0) load_batch(0);
    if (tid == 0) slot[0] = 0ull;
    __syncthreads();
    for (int bt = 0; bt < WK; ++bt) {
        const unsigned long long* src = rows + (size_t)(bt & 1) * 64 * W;
        if (wave == 3) {
            if (bt + 1 < WK) load_batch(bt + 1);
        } else if (wave == 0) {
            const int left = K - bt * 64;
            const unsigned long long valid = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
            unsigned long long alive = valid & ~slot[0];
            const unsigned long long diag = src[lane * W + bt];          // row `lane` of the diagonal block
            const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                const unsigned lo = __builtin_amdgcn_readlane(dlo, i);
                const unsigned hi = __builtin_amdgcn_readlane(dhi, i);
                if ((alive >> i) & 1ull) alive &= ~(((unsigned long long)hi << 32) | lo);
            }
            if (lane == 0) slot[1] = alive;
        }
        __syncthreads();
        if (wave < 3) {
            unsigned long long kw = slot[1];
            if (myword == bt) kept = kw;
            if (myword > bt && myword < WK) {
                while (kw) {                                              // uniform loop over the survivors of this batch
                    const int r = __ffsll((long long)kw) - 1;
                    kw &= kw - 1;
                    rem |= src[r * W + myword];
                }
            }
            if (myword == bt + 1) slot[0] = rem;
        }
        __syncthreads();
    }
    if (wave < 3 && kept) {
        const int* sidx = ws.sidx + (size_t)n * P;
        unsigned char* flag = ws.flag + (size_t)n * P;
        unsigned long long kw = kept;
        while (kw) {
            const int bb = __ffsll((long long)kw) - 1;
            kw &= kw - 1;
            flag[sidx[myword * 64 + bb]] = 1;
        }
    }
    __syncthreads();
    float* out = outputs + (size_t)n * P * ROW;
    const unsigned char* flag = ws.flag + (size_t)n * P;
    for (int i = tid; i < P; i += 256) {
        const bool k = flag[i] != 0;
        if (keep_out) keep_out[(size_t)n * P + i] = k ? 1 : 0;
        if (!k) {
            float* row = out + (size_t)i * ROW;
            for (int c = 4; c < ROW; ++c) row[c] = row[c] * 0.0f;
        }
    }
}

// ---- large-P variant (NMS_WS_SMALL_P < P <= NMS_WS_MAX_P; the SSD512 extension).  Same keys, same mask kernel, same keep set;
// the sort keeps its keys in the workspace and visits them in 16,384-key LDS residencies (bitonic steps with a stride below
// the chunk run in LDS, the few wider ones on global memory), and the scan reads mask rows from global memory / L2.
__device__ void bitonic_steps_lds(unsigned long long* keys, int base, int n, int k, int jstart) {
    for (int j = jstart; j > 0; j >>= 1) {
        for (int t = threadIdx.x; t < (n >> 1); t += NMS_THREADS) {
            const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
            const unsigned long long a = keys[i], b = keys[l];
            const bool desc = ((base + i) & k) == 0;
            if ((a < b) == desc) { keys[i] = b; keys[l] = a; }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(NMS_THREADS) void nms_sort_big_kernel(const float* __restrict__ outputs, int P, int C, NmsWs ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned long long* lk = reinterpret_cast<unsigned long long*>(lds);
    __shared__ int cnt;
    const int n = blockIdx.x, tid = threadIdx.x;
    const int ROW = 4 + C;
    const float* out = outputs + (size_t)n * P * ROW;
    unsigned long long* gk = ws.keys + (size_t)n * ws.sortcap;
    if (tid == 0) cnt = 0;
    __syncthreads();
    for (int i = tid; i < P; i += NMS_THREADS) {
        const float* row = out + (size_t)i * ROW;
        float sc = row[5];
        for (int c = 6; c < ROW; ++c) sc = fmaxf(sc, row[c]);
        ws.flag[(size_t)n * P + i] = 0;
        if (sc > 0.0f) {
            const int slot = atomicAdd(&cnt, 1);
            gk[slot] = ((unsigned long long)__float_as_uint(sc) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
        }
    }
    __syncthreads();
    const int K = cnt;
    int sortn = 64;
    while (sortn < K) sortn <<= 1;
    for (int i = K + tid; i < sortn; i += NMS_THREADS) gk[i] = 0ull;
    __syncthreads();
    const int ch = min(sortn, SORT_CHUNK);
    for (int c0 = 0; c0 < sortn; c0 += ch) {                 // every phase that fits one chunk, in one residency per chunk
        for (int i = tid; i < ch; i += NMS_THREADS) lk[i] = gk[c0 + i];
        __syncthreads();
        for (int k = 2; k <= ch; k <<= 1) bitonic_steps_lds(lk, c0, ch, k, k >> 1);
        for (int i = tid; i < ch; i += NMS_THREADS) gk[c0 + i] = lk[i];
        __syncthreads();
    }
    for (int k = ch << 1; k <= sortn; k <<= 1) {
        for (int j = k >> 1; j >= ch; j >>= 1) {             // strides that cross chunks: on global memory
            for (int t = tid; t < (sortn >> 1); t += NMS_THREADS) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                const unsigned long long a = gk[i], b = gk[l];
                const bool desc = (i & k) == 0;
                if ((a < b) == desc) { gk[i] = b; gk[l] = a; }
            }
            __syncthreads();
        }
        for (int c0 = 0; c0 < sortn; c0 += ch) {
            for (int i = tid; i < ch; i += NMS_THREADS) lk[i] = gk[c0 + i];
            __syncthreads();
            bitonic_steps_lds(lk, c0, ch, k, ch >> 1);
            for (int i = tid; i < ch; i += NMS_THREADS) gk[c0 + i] = lk[i];
            __syncthreads();
        }
    }
    if (tid == 0) ws.count[n] = K;
    for (int i = tid; i < K; i += NMS_THREADS) {
        const int idx = (int)(0xFFFFFFFFu - (unsigned)(gk[i] & 0xFFFFFFFFull));
        const float* row = out + (size_t)idx * ROW;
        ws.sidx[(size_t)n * P + i] = idx;
        ws.sbox[(size_t)n * P + i] = make_float4(row[0], row[1], row[2], row[3]);
        ws.scls[(size_t)n * P + i] = argmax_class(row, ROW);
    }
}

__global__ __launch_bounds__(256) void nms_scan_big_kernel(float* __restrict__ outputs, int P, int C, NmsWs ws, unsigned char* __restrict__ keep_out) {
    __shared__ unsigned long long rem[NMS_WS_MAX_P / 64];
    __shared__ unsigned long long surv;
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = ws.count[n];
    const int ROW = 4 + C, W = ws.W, WK = (K + 63) >> 6;
    const unsigned long long* mrow = ws.mask + (size_t)n * P * W;
    const int* sidx = ws.sidx + (size_t)n * P;
    unsigned char* flag = ws.flag + (size_t)n * P;
    for (int i = tid; i < WK; i += 256) rem[i] = 0ull;
    __syncthreads();
    for (int bt = 0; bt < WK; ++bt) {
        if (wave == 0) {
            const int left = K - bt * 64;
            const unsigned long long valid = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
            unsigned long long alive = valid & ~rem[bt];
            const unsigned long long diag = lane < left ? mrow[(size_t)(bt * 64 + lane) * W + bt] : 0ull;
            const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                const unsigned lo = __builtin_amdgcn_readlane(dlo, i);
                const unsigned hi = __builtin_amdgcn_readlane(dhi, i);
                if ((alive >> i) & 1ull) alive &= ~(((unsigned long long)hi << 32) | lo);
            }
            if (lane == 0) surv = alive;
        }
        __syncthreads();
        const unsigned long long kw0 = surv;
        if (tid < 64 && ((kw0 >> tid) & 1ull)) flag[sidx[bt * 64 + tid]] = 1;
        for (int w = bt + 1 + tid; w < WK; w += 256) {
            unsigned long long kw = kw0, acc = 0ull;
            while (kw) {
                const int r = __ffsll((long long)kw) - 1;
                kw &= kw - 1;
                acc |= mrow[(size_t)(bt * 64 + r) * W + w];
            }
            rem[w] |= acc;
        }
        __syncthreads();
    }
    float* out = outputs + (size_t)n * P * ROW;
    for (int i = tid; i < P; i += 256) {
        const bool k = flag[i] != 0;
        if (keep_out) keep_out[(size_t)n * P + i] = k ? 1 : 0;
        if (!k) {
            float* row = out + (size_t)i * ROW;
            for (int c = 4; c < ROW; ++c) row[c] = row[c] * 0.0f;
        }
    }
}

size_t nms_ws_layout(int N, int P, NmsWs* ws, unsigned char* base) {
    const int W = ((P + 63) / 64 + 1) / 2 * 2;      // even: mask rows stay 16-B aligned
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t o_count = take((size_t)N * 4), o_sidx = take((size_t)N * P * 4), o_sbox = take((size_t)N * P * 16);
    const size_t o_flag = take((size_t)N * P), o_mask = take((size_t)N * P * W * 8);
    int sortcap = 0;
    if (P > NMS_WS_SMALL_P) { sortcap = 64; while (sortcap < P) sortcap <<= 1; }
    const size_t o_keys = take((size_t)N * sortcap * 8);
    const size_t o_scls = take((size_t)N * P * 4);
    if (ws) {
        ws->scls = reinterpret_cast<int*>(base + o_scls); ws->per_class = 0;
        ws->keys = reinterpret_cast<unsigned long long*>(base + o_keys); ws->sortcap = sortcap;
        ws->count = reinterpret_cast<int*>(base + o_count); ws->sidx = reinterpret_cast<int*>(base + o_sidx);
        ws->sbox = reinterpret_cast<float4*>(base + o_sbox); ws->flag = base + o_flag;
        ws->mask = reinterpret_cast<unsigned long long*>(base + o_mask); ws->W = W;
    }
    return off;
}

}  // namespace

extern "C" {

int ssdhip_decode_score(float* outputs, const float* dboxes, int N, int P, int C, void* stream) {
    SSDHIP_REQUIRE(outputs && dboxes && N > 0 && P > 0, "decode_score: bad arguments");
    SSDHIP_REQUIRE(C >= 2 && C <= 1024, "decode_score: 2 <= C <= 1024");
    int64_t total = (int64_t)N * P;
    if (C == 21) hipLaunchKernelGGL(decode_score_kernel<21>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, outputs, dboxes, P, total);
    else hipLaunchKernelGGL(decode_score_rt_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, outputs, dboxes, P, C, total);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

int ssdhip_calc_iou(const float* t, const float* s, int N, int T, int S, int rowt, int rows, float* iou, void* stream) {
    SSDHIP_REQUIRE(t && s && iou && N > 0 && T > 0 && S > 0 && rowt >= 4 && rows >= 4, "calc_iou: bad arguments");
    SSDHIP_REQUIRE(T <= 65535 && N <= 65535, "calc_iou: T and N must be <= 65535");
    hipLaunchKernelGGL(calc_iou_kernel, dim3(cdiv(S, 256), T, N), dim3(256), 0, (hipStream_t)stream, t, s, T, S, rowt, rows, iou);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

int ssdhip_eval_match(const float* outputs, const float* gts, int N, int P, int G, int C, float iou_thresh, int32_t* cls, uint8_t* tp,
                      float* score, int32_t* gt_count, void* stream) {
    SSDHIP_REQUIRE(outputs && gts && cls && tp && N > 0 && P > 0 && G >= 0 && C >= 2, "eval_match: bad arguments");
    SSDHIP_REQUIRE(G <= 2048, "eval_match: at most 2048 ground-truth rows per image");
    hipLaunchKernelGGL(eval_match_kernel, dim3(N), dim3(1024), (size_t)(G > 0 ? G : 1) * 28, (hipStream_t)stream, outputs, gts, P, G, C,
                       iou_thresh, cls, tp, score, gt_count);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

int64_t ssdhip_nms_ws_bytes(int N, int P) {
    if (N <= 0 || P <= 0) return SSDHIP_EINVAL;
    return (int64_t)nms_ws_layout(N, P, nullptr, nullptr);
}

static int nms_ws_impl(float* outputs, int N, int P, int C, float iou_thresh, uint8_t* keep, void* ws_ptr, void* stream, int per_class) {
    SSDHIP_REQUIRE(outputs && ws_ptr && N > 0 && P > 0 && C >= 2, "nms_ws: bad arguments");
    SSDHIP_REQUIRE(P <= NMS_WS_MAX_P, "nms_ws: P <= 32768");
    SSDHIP_REQUIRE(N <= 65535, "nms_ws: N <= 65535");
    static DeviceOnce attr;
    if (const int dev = attr.pending(); dev != DeviceOnce::kDone) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(nms_scan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                NMS_LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(nms_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                128 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(nms_sort_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                SORT_CHUNK * 8) != hipSuccess) {
            ssdhip_set_error("nms_ws: cannot enable 128 KiB dynamic LDS");
            return SSDHIP_ELAUNCH;
        }
        attr.mark(dev);
    }
    NmsWs ws;
    nms_ws_layout(N, P, &ws, reinterpret_cast<unsigned char*>(ws_ptr));
    ws.per_class = per_class;
    hipStream_t st = (hipStream_t)stream;
    const int chunks = (P + 63) / 64;
    if (P <= NMS_WS_SMALL_P) {      // two 64-row batches of the bit-matrix fit 160 KiB of LDS; 192 word-pair loaders
        int sortn = 64;
        while (sortn < P) sortn <<= 1;
        hipLaunchKernelGGL(nms_sort_kernel, dim3(N), dim3(NMS_THREADS), (size_t)sortn * 8, st, outputs, P, C, ws);
        SSDHIP_CHECK_LAUNCH();
        hipLaunchKernelGGL(nms_mask_kernel, dim3((chunks + 3) / 4, chunks, N), dim3(256), 0, st, P, iou_thresh, ws);
        SSDHIP_CHECK_LAUNCH();
        hipLaunchKernelGGL(nms_scan_kernel, dim3(N), dim3(256), (size_t)2 * 64 * ws.W * 8 + 16, st, outputs, P, C, ws, keep);
        SSDHIP_CHECK_LAUNCH();
    } else {
        hipLaunchKernelGGL(nms_sort_big_kernel, dim3(N), dim3(NMS_THREADS), (size_t)SORT_CHUNK * 8, st, outputs, P, C, ws);
        SSDHIP_CHECK_LAUNCH();
        hipLaunchKernelGGL(nms_mask_kernel, dim3((chunks + 3) / 4, chunks, N), dim3(256), 0, st, P, iou_thresh, ws);
        SSDHIP_CHECK_LAUNCH();
        hipLaunchKernelGGL(nms_scan_big_kernel, dim3(N), dim3(256), 0, st, outputs, P, C, ws, keep);
        SSDHIP_CHECK_LAUNCH();
    }
    return SSDHIP_OK;
}

int ssdhip_nms_ws(float* outputs, int N, int P, int C, float iou_thresh, uint8_t* keep, void* ws_ptr, void* stream) {
    return nms_ws_impl(outputs, N, P, C, iou_thresh, keep, ws_ptr, stream, 0);
}

int ssdhip_nms_ws_per_class(float* outputs, int N, int P, int C, float iou_thresh, uint8_t* keep, void* ws_ptr, void* stream) {
    return nms_ws_impl(outputs, N, P, C, iou_thresh, keep, ws_ptr, stream, 1);
}

int ssdhip_nms(float* outputs, int N, int P, int C, float iou_thresh, uint8_t* keep, void* stream) {
    SSDHIP_REQUIRE(outputs && N > 0 && P > 0 && C >= 2, "nms: bad arguments");
    // LDS plan: idx16 (2P) + kept boxes (16P) + keep bits + misc must fit 160 KiB -> P <= 9000
    SSDHIP_REQUIRE(P <= 9000, "nms: P <= 9000 (all-LDS kept list); larger prior sets need the global-list variant");
    static DeviceOnce attr;
    if (const int dev = attr.pending(); dev != DeviceOnce::kDone) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(nms_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                NMS_LDS_BYTES) != hipSuccess) {
            ssdhip_set_error("nms: cannot enable 160 KiB dynamic LDS");
            return SSDHIP_ELAUNCH;
        }
        attr.mark(dev);
    }
    int idx_bytes = ((2 * P + 15) / 16) * 16;
    int keepbits_bytes = (((P + 31) / 32 + 3) / 4 * 4) * 4;
    int kept_cap = (NMS_LDS_BYTES - NMS_MISC_BYTES - keepbits_bytes - idx_bytes) / 16;
    SSDHIP_REQUIRE(kept_cap >= P, "nms: LDS plan does not fit");
    hipLaunchKernelGGL(nms_kernel, dim3(N), dim3(NMS_THREADS), NMS_LDS_BYTES, (hipStream_t)stream, outputs, P, C, iou_thresh, keep,
                       idx_bytes, kept_cap);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

}  // extern "C"
