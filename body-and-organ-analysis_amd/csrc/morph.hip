// Slice-wise morphology for the BCA post-processing:
//   boa_fill_holes_2d   per-slice external-contour fill of body_parts/postprocess.py:31-39
//                       (cv2.findContours(RETR_EXTERNAL) + drawContours(FILLED)): the filled set is the foreground
//                       plus every background pixel that is not 4-connected to the slice border through background
//                       (8-connected foreground / 4-connected background duality) -- union-find over the background.
//   boa_median3_inplane scipy.ndimage.median_filter(size 3 on two axes, 1 on the slice axis, mode="reflect") of
//                       tissue/subclassification.py:21-36 on int16 HU.
//   boa_mask_assign     out[mask (!)= 0] = value   (`out[filled] = label`, body_parts/postprocess.py:50).
#include <algorithm>

#include "common.h"

#define AGENT_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

__device__ __forceinline__ int uf2_find(int* L, int i) {
    int p = AGENT_LOAD(&L[i]);
    while (p != i) {
        i = p;
        p = AGENT_LOAD(&L[i]);
    }
    return i;
}

__device__ __forceinline__ void uf2_union(int* L, int a, int b) {
    while (true) {
        a = uf2_find(L, a);
        b = uf2_find(L, b);
        if (a == b) return;
        if (a < b) {
            int t = a;
            a = b;
            b = t;
        }
        int old = atomicMin(&L[a], b);
        if (old == a) return;
        a = old;
    }
}

__global__ __launch_bounds__(256) void k_bg_init(const unsigned char* __restrict__ mask, size_t n, int* __restrict__ L,
                                                 unsigned char* __restrict__ flag) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        L[i] = mask[i] ? -1 : (int)i;
        flag[i] = 0;
    }
}

// 4-connected, in-slice: merge with the left and the upper background neighbour
__global__ __launch_bounds__(256) void k_bg_merge(const unsigned char* __restrict__ mask, size_t n, int Y, int X, int* L) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || mask[i]) return;
    const int x = (int)(i % X);
    const int y = (int)((i / X) % Y);
    if (x > 0 && !mask[i - 1]) uf2_union(L, (int)i, (int)(i - 1));
    if (y > 0 && !mask[i - X]) uf2_union(L, (int)i, (int)(i - X));
}

__global__ __launch_bounds__(256) void k_bg_flag_border(const unsigned char* __restrict__ mask, size_t n, int Y, int X,
                                                        int* L, unsigned char* flag) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || mask[i]) return;
    const int x = (int)(i % X);
    const int y = (int)((i / X) % Y);
    if (x == 0 || y == 0 || x == X - 1 || y == Y - 1) flag[uf2_find(L, (int)i)] = 1;
}

__global__ __launch_bounds__(256) void k_bg_fill(const unsigned char* __restrict__ mask, size_t n, int* L,
                                                 const unsigned char* __restrict__ flag, unsigned char* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned char v = 1;
    if (!mask[i]) v = flag[uf2_find(L, (int)i)] ? 0 : 1;
    out[i] = v;
}

// Slice-wise fill as a bit-parallel flood in LDS: one workgroup per slice keeps the background and the "reached from the
// border" sets as bitmasks (Y x ceil(X / 32) words each) and alternates (a) a horizontal run fill per row -- adding the
// seed word to the background word makes the carry ripple through each run of ones that contains a seed, across word
// boundaries and in both directions (bit-reversed words) -- with (b) a vertical step `reach |= (up | down) & background`,
// until nothing changes.  Every iteration is a few thousand word operations; the global traffic is one read of the mask and
// one write of the result.  (The union-find form below needs ~75 ms for a 512^3 mask, this one well under a millisecond.)
// bit i of the result: byte i of the 32 bytes at p (16-byte aligned) is zero
__device__ __forceinline__ unsigned int zero_bits32(const unsigned char* p) {
    const uint4 a = *(const uint4*)p, b = *(const uint4*)(p + 16);
    const unsigned int w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    unsigned int r = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const unsigned int nz = (((w[k] & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w[k]) & 0x80808080u;  // bit 7 of every non-zero byte
        const unsigned int z4 = ((~nz & 0x80808080u) >> 7);                                     // 0x01 in every zero byte
        r |= ((z4 * 0x01020408u) >> 24 & 0xfu) << (4 * k);                                      // byte j -> bit j (j = 0..3)
    }
    return r;
}

// 32 output bytes (0 / 1) from the bits of v, 16-byte aligned destination
__device__ __forceinline__ void store_bits32(unsigned char* p, unsigned int v) {
    unsigned int w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const unsigned int n4 = (v >> (4 * k)) & 0xfu;
        w[k] = (n4 & 1u) | ((n4 & 2u) << 7) | ((n4 & 4u) << 14) | ((n4 & 8u) << 21);
    }
    *(uint4*)p = make_uint4(w[0], w[1], w[2], w[3]);
    *(uint4*)(p + 16) = make_uint4(w[4], w[5], w[6], w[7]);
}

__global__ __launch_bounds__(256) void k_fill_holes_bits(const unsigned char* __restrict__ mask, int Y, int X, int W,
                                                         unsigned char* __restrict__ out, int sweep) {
    extern __shared__ unsigned int fsm[];
    unsigned int* bg = fsm;                 // [Y][W]
    unsigned int* rc = fsm + (size_t)Y * W; // [Y][W]
    __shared__ int changed;
    const int tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * Y * X;
    const int nw = Y * W;
    for (int idx = tid; idx < nw; idx += 256) {
        const int y = idx / W, w = idx - y * W;
        const unsigned char* row = mask + base + (size_t)y * X + w * 32;
        const int cnt = min(32, X - w * 32);
        unsigned int b = 0;
        if (cnt == 32 && (((uintptr_t)row) & 15) == 0)
            b = zero_bits32(row);  // (the byte loop was most of the kernel's time: 1 024 dependent byte loads per thread)
        else
            for (int i = 0; i < cnt; ++i) b |= (row[i] == 0 ? 1u : 0u) << i;
        unsigned int r = 0;
        if (y == 0 || y == Y - 1) r = b;
        if (w == 0) r |= b & 1u;
        if (w == W - 1) r |= b & (1u << ((X - 1) & 31));
        bg[idx] = b;
        rc[idx] = r;
    }
    __syncthreads();
    for (;;) {
        if (tid == 0) changed = 0;
        __syncthreads();
        int ch = 0;
        for (int y = tid; y < Y; y += 256) {
            unsigned int* rr = rc + y * W;
            const unsigned int* bb = bg + y * W;
            unsigned int carry = 0;
            for (int w = 0; w < W; ++w) {          // towards higher x
                const unsigned int b = bb[w];
                unsigned int sd = rr[w] | (carry ? (b & 1u) : 0u);
                const unsigned long long sum = (unsigned long long)b + sd;
                const unsigned int nr = sd | ((b ^ (unsigned int)sum) & b);
                carry = (unsigned int)(sum >> 32);
                if (nr != rr[w]) { rr[w] = nr; ch = 1; }
            }
            carry = 0;
            for (int w = W - 1; w >= 0; --w) {     // towards lower x: the same on bit-reversed words
                const unsigned int b = __brev(bb[w]);
                unsigned int sd = __brev(rr[w]) | (carry ? (b & 1u) : 0u);
                const unsigned long long sum = (unsigned long long)b + sd;
                const unsigned int nr = __brev(sd | ((b ^ (unsigned int)sum) & b));
                carry = (unsigned int)(sum >> 32);
                if (nr != rr[w]) { rr[w] = nr; ch = 1; }
            }
        }
        __syncthreads();
        // vertical SWEEPS: thread w owns word column w (32 pixel columns at once) and carries the reached set down the whole
        // slice, then up -- a background column open to the border is flooded in one iteration however long it is.  (The first
        // version advanced one row per iteration: a 512-row slice of maze-like labels needed hundreds of iterations of three
        // barriers each, 0.7 ... 7.5 ms per mask depending on the labels; the sweep's loads do not depend on the carried word,
        // so the 2 Y steps pipeline.)
        if (!sweep) {
            for (int idx = tid; idx < nw; idx += 256) {   // vertical step (each word has one writer; neighbours are only read)
                const int y = idx / W;
                const unsigned int r = rc[idx];
                const unsigned int up = y > 0 ? rc[idx - W] : 0u, dn = y < Y - 1 ? rc[idx + W] : 0u;
                const unsigned int nr = r | ((up | dn) & bg[idx]);
                if (nr != r) { rc[idx] = nr; ch = 1; }
            }
        } else
        for (int wc = tid; wc < W; wc += 256) {
            // 32 rows at a time through registers: the LDS reads of a chunk are independent (issued back to back), only the
            // register chain carries the dependency -- a row-by-row loop pays one LDS round trip per row
            constexpr int CH = 32;
            unsigned int carry = 0;
            for (int y0 = 0; y0 < Y; y0 += CH) {
                unsigned int rr[CH], bb[CH];
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const int y = min(y0 + k, Y - 1);
                    rr[k] = rc[y * W + wc];
                    bb[k] = bg[y * W + wc];
                }
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    if (y0 + k < Y) {
                        const unsigned int nr = rr[k] | (carry & bb[k]);
                        if (nr != rr[k]) { rc[(y0 + k) * W + wc] = nr; ch = 1; }
                        carry = nr;
                    }
                }
            }
            carry = 0;
            for (int y1 = Y; y1 > 0; y1 -= CH) {
                unsigned int rr[CH], bb[CH];
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const int y = max(y1 - 1 - k, 0);
                    rr[k] = rc[y * W + wc];
                    bb[k] = bg[y * W + wc];
                }
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    if (y1 - 1 - k >= 0) {
                        const unsigned int nr = rr[k] | (carry & bb[k]);
                        if (nr != rr[k]) { rc[(y1 - 1 - k) * W + wc] = nr; ch = 1; }
                        carry = nr;
                    }
                }
            }
        }
        if (ch) changed = 1;
        __syncthreads();
        if (!changed) break;
        __syncthreads();
    }
    for (int idx = tid; idx < nw; idx += 256) {
        const int y = idx / W, w = idx - y * W;
        const unsigned int hole = bg[idx] & ~rc[idx];
        const unsigned int fg = ~bg[idx];
        unsigned char* orow = out + base + (size_t)y * X + w * 32;
        const int cnt = min(32, X - w * 32);
        const unsigned int v = hole | fg;
        if (cnt == 32 && (((uintptr_t)orow) & 15) == 0)
            store_bits32(orow, v);
        else
            for (int i = 0; i < cnt; ++i) orow[i] = (unsigned char)((v >> i) & 1u);
    }
}

extern "C" int boa_fill_holes_2d(boa_ctx* c, const uint8_t* dev_mask, int Z, int Y, int X, int32_t* dev_scratch_i32,
                                 uint8_t* dev_scratch_u8, uint8_t* dev_out) {
    BOA_REQUIRE(c && dev_mask && dev_scratch_i32 && dev_scratch_u8 && dev_out && Z > 0 && Y > 0 && X > 0,
                "boa_fill_holes_2d: bad argument");
    BOA_REQUIRE(dev_out != dev_scratch_u8 && dev_out != dev_mask, "boa_fill_holes_2d: out must not alias mask/scratch");
    const size_t n = (size_t)Z * Y * X;
    BOA_REQUIRE(n < (1ull << 31), "boa_fill_holes_2d: volume too large for int32 indices");
    const int W = (X + 31) / 32;
    const size_t lds = (size_t)Y * W * 8;
    static const bool bits_off = getenv("BOA_FILL_BITS") && atoi(getenv("BOA_FILL_BITS")) == 0;
    if (!bits_off && lds <= 150 * 1024) {
        static bool once = (hipFuncSetAttribute((const void*)k_fill_holes_bits, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256), true);
        (void)once;
        KernelTimer tb(c, BOA_K_MORPH, 0, (double)n * 2.0);
        // (measured on 154 x 512 x 512 masks, tools/fill_time.py: sweeps 7.1 / 2.4 / 0.6 ms on maze-like / noise / dense masks against 8.5 / 1.0 / 0.4 ms
        //  for the one-row step: not a win on the labels the bench produces -- kept as an experiment hook)
        static const int sweep = getenv("BOA_FILL_SWEEP") ? atoi(getenv("BOA_FILL_SWEEP")) : 0;
        hipLaunchKernelGGL(k_fill_holes_bits, dim3(Z), dim3(256), lds, c->stream, dev_mask, Y, X, W, dev_out, sweep);
        tb.stop();
        BOA_HIP_TRY(hipGetLastError());
        return BOA_OK;
    }
    const unsigned grid = (unsigned)((n + 255) / 256);
    KernelTimer t(c, BOA_K_MORPH, 0, (double)n * 12.0);
    hipLaunchKernelGGL(k_bg_init, dim3(grid), dim3(256), 0, c->stream, dev_mask, n, dev_scratch_i32, dev_scratch_u8);
    hipLaunchKernelGGL(k_bg_merge, dim3(grid), dim3(256), 0, c->stream, dev_mask, n, Y, X, dev_scratch_i32);
    hipLaunchKernelGGL(k_bg_flag_border, dim3(grid), dim3(256), 0, c->stream, dev_mask, n, Y, X, dev_scratch_i32,
                       dev_scratch_u8);
    hipLaunchKernelGGL(k_bg_fill, dim3(grid), dim3(256), 0, c->stream, dev_mask, n, dev_scratch_i32, dev_scratch_u8, dev_out);
    t.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cswap(short& a, short& b) {
    const short lo = a < b ? a : b, hi = a < b ? b : a;
    a = lo;
    b = hi;
}

// flat_axis: the axis (0 = z, 1 = y, 2 = x of the [Z][Y][X] array) with kernel size 1
__global__ __launch_bounds__(256) void k_median3_inplane(const short* __restrict__ in, int Z, int Y, int X, int flat_axis,
                                                         short* __restrict__ out) {
    const size_t n = (size_t)Z * Y * X;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int x = (int)(i % X);
    const int y = (int)((i / X) % Y);
    const int z = (int)(i / ((size_t)X * Y));
    const int dims[3] = {Z, Y, X};
    const int pos[3] = {z, y, x};
    const int a0 = flat_axis == 0 ? 1 : 0, a1 = flat_axis == 2 ? 1 : 2;
    const size_t strides[3] = {(size_t)Y * X, (size_t)X, 1};
    short v[9];
#pragma unroll
    for (int d0 = -1; d0 <= 1; ++d0)
#pragma unroll
        for (int d1 = -1; d1 <= 1; ++d1) {
            // mode="reflect" (d c b a | a b c d): for a radius-1 window the mirrored index is the clamped index
            const int p0 = min(max(pos[a0] + d0, 0), dims[a0] - 1);
            const int p1 = min(max(pos[a1] + d1, 0), dims[a1] - 1);
            v[(d0 + 1) * 3 + d1 + 1] = in[(size_t)pos[flat_axis] * strides[flat_axis] + (size_t)p0 * strides[a0] +
                                          (size_t)p1 * strides[a1]];
        }
    // 19-exchange median-of-9 network
    cswap(v[1], v[2]); cswap(v[4], v[5]); cswap(v[7], v[8]);
    cswap(v[0], v[1]); cswap(v[3], v[4]); cswap(v[6], v[7]);
    cswap(v[1], v[2]); cswap(v[4], v[5]); cswap(v[7], v[8]);
    cswap(v[0], v[3]); cswap(v[5], v[8]); cswap(v[4], v[7]);
    cswap(v[3], v[6]); cswap(v[1], v[4]); cswap(v[2], v[5]);
    cswap(v[4], v[7]); cswap(v[4], v[2]); cswap(v[6], v[4]);
    cswap(v[4], v[2]);
    out[i] = v[4];
}

extern "C" int boa_median3_inplane(boa_ctx* c, const int16_t* dev_in, int Z, int Y, int X, int flat_axis, int16_t* dev_out) {
    BOA_REQUIRE(c && dev_in && dev_out && dev_in != dev_out && Z > 0 && Y > 0 && X > 0, "boa_median3_inplane: bad argument");
    BOA_REQUIRE(flat_axis >= 0 && flat_axis <= 2, "boa_median3_inplane: flat_axis %d", flat_axis);
    const size_t n = (size_t)Z * Y * X;
    KernelTimer t(c, BOA_K_MORPH, 0, (double)n * 4.0);
    hipLaunchKernelGGL(k_median3_inplane, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, dev_in, Z, Y, X, flat_axis,
                       dev_out);
    t.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

__global__ __launch_bounds__(256) void k_mask_assign(const unsigned char* __restrict__ mask, size_t n, int invert, int value,
                                                     unsigned char* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const bool on = (mask[i] != 0) != (invert != 0);
    if (on) out[i] = (unsigned char)value;
}

extern "C" int boa_mask_assign(boa_ctx* c, const uint8_t* dev_mask, size_t n, int invert, int value, uint8_t* dev_out) {
    BOA_REQUIRE(c && dev_mask && dev_out, "boa_mask_assign: NULL argument");
    if (n == 0) return BOA_OK;
    hipLaunchKernelGGL(k_mask_assign, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, dev_mask, n, invert, value,
                       dev_out);
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

__global__ __launch_bounds__(256) void k_label_overlay(const unsigned char* __restrict__ part, size_t n,
                                                       unsigned char* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned char v = part[i];
    if (v) out[i] = v;
}

extern "C" int boa_label_overlay(boa_ctx* c, const uint8_t* dev_part, size_t n, uint8_t* dev_out) {
    BOA_REQUIRE(c && dev_part && dev_out, "boa_label_overlay: NULL argument");
    if (n == 0) return BOA_OK;
    hipLaunchKernelGGL(k_label_overlay, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, dev_part, n, dev_out);
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

// ------------------------------------------------------------------------------------------------------
// boa_copy3: strided 3-D gather copy with dtype conversion -- the device form of the index remaps around a task
// (TS/alignment.py reorientation = axis permutation + flips, TS/cropping.py crop / un-crop, the (x,y,z) <-> (z,y,x)
// view change between nibabel and nnU-Net arrays, the z-splits of TS/nnunet.py:495-505 and their recombination).
//   out[(o + out_off) in out_full] = convert(in[in_off + sum_k o_k * in_step[k]])     for o in [0, dims)
// in_step[k] (elements, may be negative) is the input stride walked by output axis k; in_off the element offset of o = 0.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void k_copy3(const TI* __restrict__ in, long long in_off, long long s0, long long s1, long long s2,
                                               int d0, int d1, int d2, TO* __restrict__ out, long long out_off, long long t0,
                                               long long t1, long long t2) {
    const size_t n = (size_t)d0 * d1 * d2;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int o0, o1, o2;
    idx3(i, d1, d2, o0, o1, o2);
    const TI v = in[in_off + o0 * s0 + o1 * s1 + o2 * s2];
    out[out_off + o0 * t0 + o1 * t1 + o2 * t2] = (TO)v;  // float -> int conversions truncate like numpy astype
}

// Transposing remaps (the input's contiguous axis is output axis A != 2, e.g. the (x,y,z) <-> (z,y,x) view change of whole
// volumes): 64 x 64 tiles of the (A, 2) plane go through LDS so that both the reads (along A) and the writes (along axis 2)
// are contiguous; the remaining output axis C is the batch dimension.  The element-wise form reads one cache line per
// element on such a remap (a 512^3 int16 transpose took ~25 ms instead of ~0.2 ms).
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void k_copy3_t(const TI* __restrict__ in, long long in_off, long long sA, long long sB, long long sC,
                                                 int dA, int dB, int dC, TO* __restrict__ out, long long out_off, long long tA,
                                                 long long tB, long long tC) {
    __shared__ TO tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int a0 = blockIdx.x * 64, b0 = blockIdx.y * 64, c = blockIdx.z;
    const TI* ip = in + in_off + (long long)c * sC;
    TO* op = out + out_off + (long long)c * tC;
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
        const int a = a0 + tx, b = b0 + ty + 4 * k;
        if (a < dA && b < dB) tile[ty + 4 * k][tx] = (TO)ip[a * sA + b * sB];
    }
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
        const int b = b0 + tx, a = a0 + ty + 4 * k;
        if (a < dA && b < dB) op[a * tA + b * tB] = tile[tx][ty + 4 * k];
    }
}

template <typename TI>
static int copy3_out(boa_ctx* c, const void* in, long long in_off, const long long s[3], const int d[3], void* out, int out_dtype,
                     long long out_off, const long long t[3]) {
    const size_t n = (size_t)d[0] * d[1] * d[2];
    const unsigned grid = (unsigned)((n + 255) / 256);
    auto iabs = [](long long v) { return v < 0 ? -v : v; };
    // input-contiguous output axis A (0 or 1) while the output is contiguous along axis 2: tiled transpose
    int A = -1;
    if (iabs(t[2]) == 1 && iabs(s[2]) != 1 && d[2] >= 16) {
        if (iabs(s[1]) == 1 && d[1] >= 16) A = 1;
        else if (iabs(s[0]) == 1 && d[0] >= 16) A = 0;
    }
    if (A >= 0 && d[1 - A] <= 65535) {
        const int Cx = 1 - A;
        const dim3 g((unsigned)((d[A] + 63) / 64), (unsigned)((d[2] + 63) / 64), (unsigned)d[Cx]);
        if (g.y <= 65535) {
#define LAUNCH_T(TO) hipLaunchKernelGGL((k_copy3_t<TI, TO>), g, dim3(256), 0, c->stream, (const TI*)in, in_off, s[A], s[2], s[Cx], d[A], d[2], \
                                        d[Cx], (TO*)out, out_off, t[A], t[2], t[Cx])
            switch (out_dtype) {
                case 0: LAUNCH_T(uint8_t); break;
                case 1: LAUNCH_T(int16_t); break;
                case 2: LAUNCH_T(int32_t); break;
                case 3: LAUNCH_T(float); break;
                case 4: LAUNCH_T(double); break;
                default: boa_set_error("boa_copy3: out dtype %d", out_dtype); return BOA_EINVAL;
            }
#undef LAUNCH_T
            return BOA_OK;
        }
    }
#define LAUNCH(TO) hipLaunchKernelGGL((k_copy3<TI, TO>), dim3(grid), dim3(256), 0, c->stream, (const TI*)in, in_off, s[0], s[1], s[2], \
                                      d[0], d[1], d[2], (TO*)out, out_off, t[0], t[1], t[2])
    switch (out_dtype) {
        case 0: LAUNCH(uint8_t); break;
        case 1: LAUNCH(int16_t); break;
        case 2: LAUNCH(int32_t); break;
        case 3: LAUNCH(float); break;
        case 4: LAUNCH(double); break;
        default: boa_set_error("boa_copy3: out dtype %d", out_dtype); return BOA_EINVAL;
    }
#undef LAUNCH
    return BOA_OK;
}

// dtype codes: 0 uint8, 1 int16, 2 int32, 3 float32, 4 float64
extern "C" int boa_copy3(boa_ctx* c, const void* dev_in, int in_dtype, long long in_off, const long long in_step[3],
                         const int dims[3], void* dev_out, int out_dtype, long long out_off, const long long out_step[3]) {
    BOA_REQUIRE(c && dev_in && dev_out && in_step && dims && out_step, "boa_copy3: NULL argument");
    BOA_REQUIRE(dims[0] >= 0 && dims[1] >= 0 && dims[2] >= 0, "boa_copy3: negative dims");
    if ((size_t)dims[0] * dims[1] * dims[2] == 0) return BOA_OK;
    static const int isz[5] = {1, 2, 4, 4, 8};
    KernelTimer t(c, BOA_K_COPY, 0, (double)dims[0] * dims[1] * dims[2] * ((in_dtype >= 0 && in_dtype < 5 ? isz[in_dtype] : 0) +
                                                                          (out_dtype >= 0 && out_dtype < 5 ? isz[out_dtype] : 0)));
    int rc;
    switch (in_dtype) {
        case 0: rc = copy3_out<uint8_t>(c, dev_in, in_off, in_step, dims, dev_out, out_dtype, out_off, out_step); break;
        case 1: rc = copy3_out<int16_t>(c, dev_in, in_off, in_step, dims, dev_out, out_dtype, out_off, out_step); break;
        case 2: rc = copy3_out<int32_t>(c, dev_in, in_off, in_step, dims, dev_out, out_dtype, out_off, out_step); break;
        case 3: rc = copy3_out<float>(c, dev_in, in_off, in_step, dims, dev_out, out_dtype, out_off, out_step); break;
        case 4: rc = copy3_out<double>(c, dev_in, in_off, in_step, dims, dev_out, out_dtype, out_off, out_step); break;
        default: boa_set_error("boa_copy3: in dtype %d", in_dtype); rc = BOA_EINVAL;
    }
    t.stop();
    if (rc) return rc;
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}

// bounding box of `data != 0` (crop_to_nonzero, NN/preprocessing/cropping/cropping.py:6-29): per axis [min, max + 1),
// or [0, dim) when everything is zero.  host_bbox: int[6] = {lo0, hi0, lo1, hi1, lo2, hi2}.  Synchronous.
template <typename T>
__global__ __launch_bounds__(256) void k_nonzero_bbox(const T* __restrict__ in, int d0, int d1, int d2, int* __restrict__ bb, int vec) {
    // one wave per row of the contiguous axis: the (axis 0, axis 1) indices come from the row number once per row (the first
    // version decomposed every non-zero voxel's linear index with 64-bit divisions: 1.2 ms per 512^3 mask, 115 GB/s)
    int lo[3] = {1 << 30, 1 << 30, 1 << 30}, hi[3] = {-1, -1, -1};
    const unsigned rows = (unsigned)d0 * (unsigned)d1;
    const unsigned lane = threadIdx.x & 63, nw = gridDim.x * 4;
    for (unsigned r = blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += nw) {
        const T* p = in + (size_t)r * d2;
        int lx = 1 << 30, hx = -1;
        if (vec) {
            // 16 bytes per lane and load (byte loads of a uint8 mask ran at 60 GB/s: 2.2 ms per 512^3 volume)
            constexpr int E = 16 / (int)sizeof(T);
            const int nchunks = d2 / E;
            for (int c = (int)lane; c < nchunks; c += 64) {
                union {
                    uint4 u;
                    T e[E];
                } v;
                v.u = ((const uint4*)p)[c];
                if ((v.u.x | v.u.y | v.u.z | v.u.w) != 0u) {   // (float: -0.0 has a sign bit but compares equal to 0: checked per element below)
#pragma unroll
                    for (int k = 0; k < E; ++k)
                        if (v.e[k] != (T)0) {
                            lx = min(lx, c * E + k);
                            hx = max(hx, c * E + k);
                        }
                }
            }
        } else {
            for (int x = (int)lane; x < d2; x += 64)
                if (p[x] != (T)0) {
                    lx = min(lx, x);
                    hx = max(hx, x);
                }
        }
        if (hx >= 0) {
            const int o0 = (int)(r / (unsigned)d1), o1 = (int)(r - (unsigned)o0 * (unsigned)d1);
            lo[0] = min(lo[0], o0); hi[0] = max(hi[0], o0);
            lo[1] = min(lo[1], o1); hi[1] = max(hi[1], o1);
            lo[2] = min(lo[2], lx); hi[2] = max(hi[2], hx);
        }
    }
    // wave reduction, then the block's four waves through LDS: ONE set of six atomics per block (one per wave meant ~200 000
    // atomics on the same six words: 2.2 ms per call whatever the element size -- the kernel was atomic-bound, not memory-bound)
    __shared__ int red[4][6];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            lo[a] = min(lo[a], __shfl_xor(lo[a], m));
            hi[a] = max(hi[a], __shfl_xor(hi[a], m));
        }
        if ((threadIdx.x & 63) == 0) {
            red[threadIdx.x >> 6][2 * a] = lo[a];
            red[threadIdx.x >> 6][2 * a + 1] = hi[a];
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = (int)threadIdx.x;
        int l = red[0][2 * a], h = red[0][2 * a + 1];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            l = min(l, red[w][2 * a]);
            h = max(h, red[w][2 * a + 1]);
        }
        if (h >= 0) {
            atomicMin(&bb[2 * a], l);
            atomicMax(&bb[2 * a + 1], h);
        }
    }
}

extern "C" int boa_nonzero_bbox(boa_ctx* c, const void* dev_in, int dtype, const int dims[3], int host_bbox[6]) {
    BOA_REQUIRE(c && dev_in && dims && host_bbox, "boa_nonzero_bbox: NULL argument");
    int* d_bb = nullptr;
    BOA_TRY(boa_malloc(c, 6 * sizeof(int), (void**)&d_bb));   // pooled (hipMalloc / hipFree synchronise the device)
    const int init[6] = {1 << 30, -1, 1 << 30, -1, 1 << 30, -1};
    BOA_HIP_TRY(hipMemcpyAsync(d_bb, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
    const size_t n = (size_t)dims[0] * dims[1] * dims[2];
    const unsigned grid = (unsigned)std::min<size_t>(((size_t)dims[0] * dims[1] + 3) / 4, (size_t)c->cu_count * 8);
    c->prof_break = true;
    const int esz = dtype == 0 ? 1 : (dtype == 1 ? 2 : 4);
    const int vec = ((size_t)dev_in % 16 == 0 && ((size_t)dims[2] * esz) % 16 == 0) ? 1 : 0;   // rows start on 16-byte boundaries
    if (n) {
        switch (dtype) {
            case 0: hipLaunchKernelGGL(k_nonzero_bbox<uint8_t>, dim3(grid), dim3(256), 0, c->stream, (const uint8_t*)dev_in, dims[0], dims[1], dims[2], d_bb, vec); break;
            case 1: hipLaunchKernelGGL(k_nonzero_bbox<int16_t>, dim3(grid), dim3(256), 0, c->stream, (const int16_t*)dev_in, dims[0], dims[1], dims[2], d_bb, vec); break;
            case 2: hipLaunchKernelGGL(k_nonzero_bbox<int32_t>, dim3(grid), dim3(256), 0, c->stream, (const int32_t*)dev_in, dims[0], dims[1], dims[2], d_bb, vec); break;
            case 3: hipLaunchKernelGGL(k_nonzero_bbox<float>, dim3(grid), dim3(256), 0, c->stream, (const float*)dev_in, dims[0], dims[1], dims[2], d_bb, vec); break;
            default: boa_free(c, d_bb); boa_set_error("boa_nonzero_bbox: dtype %d (0 uint8, 1 int16, 2 int32, 3 float32)", dtype); return BOA_EINVAL;
        }
    }
    int bb[6];
    hipError_t e = hipMemcpyAsync(bb, d_bb, sizeof(bb), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    boa_free(c, d_bb);
    BOA_HIP_TRY(e);
    for (int a = 0; a < 3; ++a) {
        const bool any = bb[2 * a + 1] >= 0;
        host_bbox[2 * a] = any ? bb[2 * a] : 0;
        host_bbox[2 * a + 1] = any ? bb[2 * a + 1] + 1 : dims[a];
    }
    return BOA_OK;
}


// ------------------------------------------------------------------------------------------------------
// remove_outside_of_mask (TS/postprocessing.py:101-131): mask = scipy.ndimage.binary_dilation(mask, iterations=addon)
// with the default structuring element (the 6-neighbour cross, border_value 0), then seg[mask == 0] = 0.
__global__ __launch_bounds__(256) void k_dilate_cross(const unsigned char* __restrict__ in, int Z, int Y, int X, unsigned char* __restrict__ out) {
    const size_t n = (size_t)Z * Y * X;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int x = (int)(i % X), y = (int)((i / X) % Y), z = (int)(i / ((size_t)X * Y));
    const size_t sy = X, sz = (size_t)X * Y;
    bool v = in[i] != 0;
    if (!v) {
        v = (x > 0 && in[i - 1]) || (x < X - 1 && in[i + 1]) || (y > 0 && in[i - sy]) || (y < Y - 1 && in[i + sy]) ||
            (z > 0 && in[i - sz]) || (z < Z - 1 && in[i + sz]);
    }
    out[i] = v ? 1 : 0;
}

extern "C" int boa_binary_dilate_cross(boa_ctx* c, const uint8_t* dev_mask, uint8_t* dev_out, uint8_t* dev_tmp, int Z, int Y, int X,
                                       int iterations) {
    BOA_REQUIRE(c && dev_mask && dev_out && dev_tmp && Z > 0 && Y > 0 && X > 0 && iterations >= 0, "boa_binary_dilate_cross: bad argument");
    BOA_REQUIRE(dev_out != dev_tmp && dev_out != dev_mask && dev_tmp != dev_mask, "boa_binary_dilate_cross: buffers must not alias");
    const size_t n = (size_t)Z * Y * X;
    const unsigned grid = (unsigned)((n + 255) / 256);
    KernelTimer t(c, BOA_K_MORPH, 0, (double)n * 2.0 * std::max(iterations, 1));
    if (iterations == 0) {
        // scipy: iterations < 1 repeats until nothing changes; the reference only passes int(mm / mean spacing) >= 0 and a
        // 0 there means "until convergence" = everything reachable: the whole volume if the mask is not empty
        BOA_HIP_TRY(hipMemcpyAsync(dev_out, dev_mask, n, hipMemcpyDeviceToDevice, c->stream));
        boa_set_error("boa_binary_dilate_cross: iterations == 0 (dilate until convergence) is not supported");
        return BOA_EINVAL;
    }
    // ping-pong so that the last iteration lands in dev_out
    const unsigned char* src = dev_mask;
    for (int it = 0; it < iterations; ++it) {
        unsigned char* dst = ((iterations - it) % 2 == 1) ? dev_out : dev_tmp;
        hipLaunchKernelGGL(k_dilate_cross, dim3(grid), dim3(256), 0, c->stream, src, Z, Y, X, dst);
        src = dst;
    }
    t.stop();
    BOA_HIP_TRY(hipGetLastError());
    return BOA_OK;
}
