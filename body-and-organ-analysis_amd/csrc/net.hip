// PlainConvUNet on device + the sliding-window tile loop
// (NN/inference/predict_from_raw_data.py:543,560-631; architecture per NN/utilities/plans_handling/plans_handler.py:59-92).
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "conv.h"

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct ConvLayer {
    ConvGeom g{};
    ConvTile t{};
    int Cin0 = 0, Cin1 = 0;  // channels of the two concatenated sources (Cin1 = 0: single source)
    bool first = false;      // stage-0 conv-0: fp32 VALU kernel reading the volume
    __half* wpk = nullptr;   // MFMA layers
    float* wfirst = nullptr; // first layer [Cin][taps][Cout]
    float* w32 = nullptr;    // fp32 mode: [taps][Cin][Cout]
    float* out32 = nullptr;  // fp32 mode: raw conv output [N][vox][Cout]; split-precision mode: octet planes [N][Cout/8][vox][8]
    float wscale = 1.f;      // split-precision mode: power-of-two scale of the packed weights (per weight set)
    float *bias = nullptr, *gamma = nullptr, *beta = nullptr;
    __half* out = nullptr;
    float* partials = nullptr;
    float* ss = nullptr;
    unsigned* ss16 = nullptr;
    int nblk = 0;
    size_t w_elems = 0;  // fp32 elements of W in the blob
};

struct UpLayer {
    int Cin = 0, Cout = 0;
    int s[3] = {1, 1, 1};
    int din[3] = {0, 0, 0};
    __half* wpk = nullptr;
    float* w32 = nullptr;    // fp32 mode: [taps][Cin][Cout]
    float* bias = nullptr;
    __half* out = nullptr;
    float* out32 = nullptr;  // fp32 / split-precision mode
    float wscale = 1.f;      // split-precision mode
};

}  // namespace

// The tile shapes of a network are chosen for a REFERENCE tile batch, not for the batch of a call or the net's max_batch: the shape
// decides how the InstanceNorm partial sums are grouped, and results must not depend on how many tiles share a launch.
// 16 = the product's default tile batch (with 8, the value of rounds 1-2, the 8^3 layers got half-size tiles -- 2 x 10 workgroups
// per sample -- which at the batches actually run (16, 25) only doubled the weight streaming: 116 -> 82, 200 -> 134, 111 -> 73 us
// per 25 tiles for the three 8^3 convs).  BOA_TILE_REF_N: experiment hook.
static int tile_ref_batch() {
    static const int v = getenv("BOA_TILE_REF_N") ? std::max(1, atoi(getenv("BOA_TILE_REF_N"))) : 16;
    return v;
}

struct boa_net {
    boa_ctx* ctx = nullptr;
    boa_net_desc d{};
    int maxN = 1;
    int precision = 0;         // 0: fp16 storage + f16 MFMA (production); 1: fp32 reference mode (net_f32.hip); 2: split-precision
                               // mode (fp32 storage, hi / lo fp16 operands on the matrix cores: k_conv_ws<X3>, net_x3.hip)
    int mirror_mask = 0;       // test-time mirroring axes (bit a = array axis a), predict_from_raw_data.py:541-557
    float* mirror_tmp = nullptr;  // [maxN][C][P] fp32 logits of one mirror variant
    float* mirror_sum = nullptr;  // [maxN][C][P] running sum / mean
    float* tiles32 = nullptr;  // fp32 mode: gathered input tiles [N][P][Cin]
    std::vector<std::vector<ConvLayer>> enc;  // [stage][conv]
    std::vector<UpLayer> up;                  // decoder order (deepest first)
    std::vector<std::vector<ConvLayer>> dec;  // [d][conv]
    float *head_w = nullptr, *head_b = nullptr;
    int* dev_origins = nullptr;
    float* first_padded = nullptr;  // zero-padded fp32 gather buffer of the first conv
    std::vector<void*> allocs;
    // Activation buffers (one per layer, ~1 GB per tile at 128^3: 25 GB at tile batch 25) live in ONE arena per context that all
    // of its networks share: the networks of a context run one after the other on its stream and every forward overwrites a
    // layer's buffer before reading it, so seven resident networks need the largest network's activations once, not seven times
    // (175 GB -> 25 GB at the bench's batch; what persists across forwards -- statistics partials, (scale, shift) tables, weight
    // arenas, the gather head's stash -- stays outside).  A layer records its offset; pointers are (re)bound whenever the arena
    // has been re-allocated for a larger network (boa_ctx::act_gen).
    struct ActSlot {
        void** where;
        size_t offset;
    };
    std::vector<ActSlot> act_slots;
    size_t act_need = 0;
    unsigned long long act_gen_seen = 0;
    int dims[BOA_MAX_STAGES][3];
    // packed weight sets (one device arena each), cached per host blob: switching folds is a pointer swap, not a re-pack
    struct WeightSet {
        const float* key;
        size_t n;
        unsigned long long sample_hash;  // FNV-1a over ~4096 evenly spaced floats: guards against a recycled host address
        unsigned char* arena;
        std::vector<float> scales;       // split-precision mode: weight scale of every conv / transposed conv, blob order
    };
    std::vector<WeightSet> wsets;
};

static int net_alloc(boa_net* net, size_t bytes, void** out) {
    BOA_TRY(boa_malloc_raw(net->ctx, bytes, out));   // (long-lived: not through the caching allocator; freed with hipFree)
    net->allocs.push_back(*out);
    return BOA_OK;
}

// an activation buffer: a slice of the context's shared arena (bound by net_bind_arena)
static int net_alloc_act(boa_net* net, size_t bytes, void** out) {
    static const bool own = getenv("BOA_NO_ACT_ARENA") != nullptr;   // experiment hook: private buffers as in rounds 1-3
    if (own) return net_alloc(net, bytes, out);
    *out = nullptr;
    net->act_slots.push_back({out, net->act_need});
    net->act_need += (bytes + 255) & ~(size_t)255;
    return BOA_OK;
}

// make the arena large enough for this network (re-allocating it if another, smaller network sized it) and point the layers at it
static int net_bind_arena(boa_net* net) {
    boa_ctx* c = net->ctx;
    if (net->act_slots.empty()) return BOA_OK;
    if (c->act_bytes < net->act_need) {
        BOA_HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->act_arena) hipFree(c->act_arena);
        c->act_arena = nullptr;
        c->act_bytes = 0;
        BOA_TRY(boa_malloc_raw(c, net->act_need, &c->act_arena));
        c->act_bytes = net->act_need;
        c->act_gen++;
    }
    if (net->act_gen_seen != c->act_gen) {
        for (auto& sl : net->act_slots) *sl.where = (unsigned char*)c->act_arena + sl.offset;
        net->act_gen_seen = c->act_gen;
    }
    return BOA_OK;
}

static bool desc_ok(const boa_net_desc* d) {
    if (!d || d->n_stages < 2 || d->n_stages > BOA_MAX_STAGES || d->in_channels < 1 || d->num_classes < 1) return false;
    for (int s = 0; s < d->n_stages; ++s) {
        if (d->features[s] <= 0 || d->n_conv_enc[s] < 1) return false;
        for (int a = 0; a < 3; ++a)
            if ((d->kernel[s][a] != 1 && d->kernel[s][a] != 3) || d->stride[s][a] < 1 || d->stride[s][a] > 2) return false;
    }
    for (int s = 0; s < d->n_stages - 1; ++s)
        if (d->n_conv_dec[s] < 1) return false;
    return true;
}

extern "C" size_t boa_net_weight_count(const boa_net_desc* d) {
    if (!desc_ok(d)) return 0;
    size_t n = 0;
    int cin = d->in_channels;
    for (int s = 0; s < d->n_stages; ++s) {
        int taps = d->kernel[s][0] * d->kernel[s][1] * d->kernel[s][2];
        for (int i = 0; i < d->n_conv_enc[s]; ++i) {
            n += (size_t)d->features[s] * cin * taps + 3 * (size_t)d->features[s];
            cin = d->features[s];
        }
    }
    for (int k = 0; k < d->n_stages - 1; ++k) {
        int sb = d->n_stages - 1 - k;  // stage below
        int below = d->features[sb], skip = d->features[sb - 1];
        int st = d->stride[sb][0] * d->stride[sb][1] * d->stride[sb][2];
        n += (size_t)below * skip * st + skip;
        int taps = d->kernel[sb - 1][0] * d->kernel[sb - 1][1] * d->kernel[sb - 1][2];
        int ci = 2 * skip;
        for (int i = 0; i < d->n_conv_dec[k]; ++i) {
            n += (size_t)skip * ci * taps + 3 * (size_t)skip;
            ci = skip;
        }
    }
    n += (size_t)d->num_classes * d->features[0] + d->num_classes;
    return n;
}

static int setup_conv(boa_net* net, ConvLayer& L, int N, const int din[3], int cin0, int cin1, int cout, const int k[3],
                      const int s[3], bool first) {
    L.first = first;
    L.Cin0 = cin0;
    L.Cin1 = cin1;
    L.g.N = N;
    L.g.Di = din[0]; L.g.Hi = din[1]; L.g.Wi = din[2];
    L.g.Cout = cout;
    L.g.Cin = cin0 + cin1;
    int dout[3];
    for (int a = 0; a < 3; ++a) {
        L.g.k[a] = k[a];
        L.g.s[a] = s[a];
        int p = (k[a] - 1) / 2;
        dout[a] = (din[a] + 2 * p - k[a]) / s[a] + 1;
    }
    L.g.Do = dout[0]; L.g.Ho = dout[1]; L.g.Wo = dout[2];
    const int taps = k[0] * k[1] * k[2];
    L.w_elems = (size_t)cout * (cin0 + cin1) * taps;
    if (net->precision == 1) {
        BOA_REQUIRE(cout % 32 == 0, "conv %d+%d -> %d: Cout must be a multiple of 32", cin0, cin1, cout);
        size_t vox32 = (size_t)dout[0] * dout[1] * dout[2];
        BOA_TRY(net_alloc_act(net, (size_t)N * vox32 * cout * sizeof(float), (void**)&L.out32));
        BOA_TRY(net_alloc(net, (size_t)N * cout * 2 * sizeof(float), (void**)&L.ss));
        return BOA_OK;
    }
    if (net->precision == 2) {
        size_t vox2 = (size_t)dout[0] * dout[1] * dout[2];
        if (first) {
            BOA_REQUIRE(s[0] == 1 && s[1] == 1 && s[2] == 1, "first conv must have stride 1");
            BOA_REQUIRE(cout % 32 == 0 && cin0 >= 1 && cin0 <= 4, "first conv %d -> %d unsupported", cin0, cout);
            L.nblk = conv_first_nblk(dout, net->ctx->cu_count);
        } else {
            BOA_REQUIRE((cin0 % 8) == 0 && (cin1 % 8) == 0 && (cout % 32) == 0,
                        "conv %d+%d -> %d: channel counts must be multiples of 8 (in) / 32 (out)", cin0, cin1, cout);
            ConvGeom gref = L.g;
            gref.N = tile_ref_batch();
            BOA_REQUIRE(choose_conv_tile(gref, net->ctx->cu_count, &L.t, true), "no split-precision tile configuration fits conv %dx%dx%d k=%dx%dx%d",
                        din[0], din[1], din[2], k[0], k[1], k[2]);
            L.nblk = conv_nblk(L.t, net->ctx->cu_count, cout);
        }
        BOA_TRY(net_alloc_act(net, (size_t)N * vox2 * cout * sizeof(float), (void**)&L.out32));
        BOA_TRY(net_alloc(net, (size_t)N * cout * 2 * L.nblk * sizeof(float), (void**)&L.partials));
        BOA_HIP_TRY(hipMemsetAsync(L.partials, 0, (size_t)N * cout * 2 * L.nblk * sizeof(float), net->ctx->stream));
        BOA_TRY(net_alloc(net, (size_t)N * cout * 2 * sizeof(float), (void**)&L.ss));
        return BOA_OK;
    }
    if (first) {
        BOA_REQUIRE(s[0] == 1 && s[1] == 1 && s[2] == 1, "first conv must have stride 1");
        L.nblk = conv_first_nblk(dout, net->ctx->cu_count);
    } else {
        BOA_REQUIRE((cin0 % 16) == 0 && (cin1 % 16) == 0 && (cout % 32) == 0,
                    "conv %d+%d -> %d: channel counts must be multiples of 16 (in) / 32 (out)", cin0, cin1, cout);
        // the tile shape fixes the fp32 summation order inside the conv and the grouping of the InstanceNorm partial sums:
        // it is chosen for a nominal batch (tile_ref_batch), never for the actual max_batch, so that a tile's result does not
        // depend on the batch size the network was created with
        ConvGeom gref = L.g;
        gref.N = tile_ref_batch();
        BOA_REQUIRE(choose_conv_tile(gref, net->ctx->cu_count, &L.t), "no tile configuration fits conv %dx%dx%d", din[0],
                    din[1], din[2]);
        L.nblk = conv_nblk(L.t, net->ctx->cu_count, cout);
    }
    size_t vox = (size_t)dout[0] * dout[1] * dout[2];
    BOA_TRY(net_alloc_act(net, (size_t)N * vox * cout * sizeof(__half), (void**)&L.out));
    BOA_TRY(net_alloc(net, (size_t)N * cout * 2 * L.nblk * sizeof(float), (void**)&L.partials));
    BOA_HIP_TRY(hipMemsetAsync(L.partials, 0, (size_t)N * cout * 2 * L.nblk * sizeof(float), net->ctx->stream));
    BOA_TRY(net_alloc(net, (size_t)N * cout * 2 * sizeof(float), (void**)&L.ss));
    BOA_TRY(net_alloc(net, (size_t)N * cout * sizeof(unsigned), (void**)&L.ss16));
    return BOA_OK;
}

// Device layout of one weight set: every tensor of the blob, in blob order, at a 256-byte aligned offset of one arena.
// `visit(piece kind, layer pointers..., byte size)` is called in blob order; used both to size / fill the arena and
// to point the layers at it.
static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

template <typename F>
static void for_each_weight_piece(boa_net* net, F&& f) {
    auto conv = [&](ConvLayer& L) {
        const int cin = L.Cin0 + L.Cin1, cout = L.g.Cout;
        if (net->precision == 1)
            f(9, &L, nullptr, L.w_elems * sizeof(float));
        else if (net->precision == 2 && !L.first)
            f(11, &L, nullptr, conv_wpk_halves_x3(cin, cout, L.g.k) * sizeof(__half));
        else
        f(L.first ? 0 : 1, &L, nullptr, L.first ? L.w_elems * sizeof(float) : conv_wpk_halves(cin, cout, L.g.k) * sizeof(__half));
        f(2, &L, nullptr, cout * sizeof(float));  // bias
        f(3, &L, nullptr, cout * sizeof(float));  // gamma
        f(4, &L, nullptr, cout * sizeof(float));  // beta
    };
    for (auto& st : net->enc)
        for (auto& L : st) conv(L);
    for (size_t k = 0; k < net->up.size(); ++k) {
        UpLayer& U = net->up[k];
        if (net->precision == 1)
            f(10, nullptr, &U, (size_t)U.Cin * U.Cout * U.s[0] * U.s[1] * U.s[2] * sizeof(float));
        else if (net->precision == 2)
            f(12, nullptr, &U, convt_wpk_halves_x3(U.Cin, U.Cout, U.s) * sizeof(__half));
        else
            f(5, nullptr, &U, convt_wpk_halves(U.Cin, U.Cout, U.s) * sizeof(__half));
        f(6, nullptr, &U, U.Cout * sizeof(float));
        for (auto& L : net->dec[k]) conv(L);
    }
    f(7, nullptr, nullptr, (size_t)net->d.num_classes * net->d.features[0] * sizeof(float));
    f(8, nullptr, nullptr, net->d.num_classes * sizeof(float));
}

static void point_layers_at(boa_net* net, unsigned char* arena, const std::vector<float>& scales) {
    size_t off = 0, si = 0;
    for_each_weight_piece(net, [&](int kind, ConvLayer* L, UpLayer* U, size_t bytes) {
        void* p = arena + off;
        switch (kind) {
            case 11: L->wpk = (__half*)p; L->wscale = scales[si++]; break;
            case 12: U->wpk = (__half*)p; U->wscale = scales[si++]; break;
            case 0: L->wfirst = (float*)p; break;
            case 1: L->wpk = (__half*)p; break;
            case 2: L->bias = (float*)p; break;
            case 3: L->gamma = (float*)p; break;
            case 4: L->beta = (float*)p; break;
            case 5: U->wpk = (__half*)p; break;
            case 6: U->bias = (float*)p; break;
            case 7: net->head_w = (float*)p; break;
            case 9: L->w32 = (float*)p; break;
            case 10: U->w32 = (float*)p; break;
            default: net->head_b = (float*)p; break;
        }
        off += align256(bytes);
    });
}

extern "C" int boa_net_load_weights(boa_net* net, const float* w, size_t n_floats) {
    BOA_REQUIRE(net && w, "boa_net_load_weights: NULL argument");
    size_t expect = boa_net_weight_count(&net->d);
    BOA_REQUIRE(n_floats == expect, "weight blob has %zu floats, geometry needs %zu", n_floats, expect);
    boa_ctx* c = net->ctx;
    // cached set of the same host blob (fold switching in predict_logits_from_preprocessed_data, :483-489)?
    unsigned long long hsh = 1469598103934665603ull;
    {
        const size_t step = n_floats > 4096 ? n_floats / 4096 : 1;
        for (size_t i = 0; i < n_floats; i += step) {
            unsigned u;
            memcpy(&u, w + i, 4);
            hsh = (hsh ^ u) * 1099511628211ull;
        }
    }
    for (auto& ws : net->wsets)
        if (ws.key == w && ws.n == n_floats && ws.sample_hash == hsh) {
            point_layers_at(net, ws.arena, ws.scales);  // (host-side pointers of later launches only: queued work keeps its own)
            return BOA_OK;
        }
    BOA_HIP_TRY(hipStreamSynchronize(c->stream));
    size_t total = 0;
    for_each_weight_piece(net, [&](int, ConvLayer*, UpLayer*, size_t bytes) { total += align256(bytes); });
    std::vector<unsigned char> stage(total, 0);
    std::vector<float> scales;
    const float* p = w;
    size_t off = 0;
    for_each_weight_piece(net, [&](int kind, ConvLayer* L, UpLayer* U, size_t bytes) {
        unsigned char* dst = stage.data() + off;
        switch (kind) {
            case 0: {  // first conv: [cout][cin][taps] -> [cin][taps][cout] fp32
                const int cin = L->Cin0 + L->Cin1, cout = L->g.Cout, taps = L->g.k[0] * L->g.k[1] * L->g.k[2];
                float* wf = (float*)dst;
                for (int co = 0; co < cout; ++co)
                    for (int ci = 0; ci < cin; ++ci)
                        for (int t = 0; t < taps; ++t) wf[((size_t)ci * taps + t) * cout + co] = p[((size_t)co * cin + ci) * taps + t];
                p += L->w_elems;
                break;
            }
            case 1:
                pack_conv_weights(p, L->Cin0 + L->Cin1, L->g.Cout, L->g.k, (__half*)dst);
                p += L->w_elems;
                break;
            case 5:
                pack_convt_weights(p, U->Cin, U->Cout, U->s, (__half*)dst);
                p += (size_t)U->Cin * U->Cout * U->s[0] * U->s[1] * U->s[2];
                break;
            case 11: {  // split-precision conv: hi / lo fp16 parts of w * 2^e
                const float sc = x3_weight_scale(p, L->w_elems);
                scales.push_back(sc);
                pack_conv_weights_x3(p, L->Cin0 + L->Cin1, L->g.Cout, L->g.k, sc, (__half*)dst);
                p += L->w_elems;
                break;
            }
            case 12: {
                const size_t ne = (size_t)U->Cin * U->Cout * U->s[0] * U->s[1] * U->s[2];
                const float sc = x3_weight_scale(p, ne);
                scales.push_back(sc);
                pack_convt_weights_x3(p, U->Cin, U->Cout, U->s, sc, (__half*)dst);
                p += ne;
                break;
            }
            case 9: {  // fp32 mode conv: [cout][cin][taps] -> [tap][cin][cout]
                const int cin = L->Cin0 + L->Cin1, cout = L->g.Cout, taps = L->g.k[0] * L->g.k[1] * L->g.k[2];
                float* wf = (float*)dst;
                for (int co = 0; co < cout; ++co)
                    for (int ci = 0; ci < cin; ++ci)
                        for (int t = 0; t < taps; ++t) wf[((size_t)t * cin + ci) * cout + co] = p[((size_t)co * cin + ci) * taps + t];
                p += L->w_elems;
                break;
            }
            case 10: {  // fp32 mode convT: [cin][cout][taps] -> [tap][cin][cout]
                const int taps = U->s[0] * U->s[1] * U->s[2];
                float* wf = (float*)dst;
                for (int ci = 0; ci < U->Cin; ++ci)
                    for (int co = 0; co < U->Cout; ++co)
                        for (int t = 0; t < taps; ++t) wf[((size_t)t * U->Cin + ci) * U->Cout + co] = p[((size_t)ci * U->Cout + co) * taps + t];
                p += (size_t)U->Cin * U->Cout * taps;
                break;
            }
            default:  // fp32 vectors / head matrix, copied as they are
                memcpy(dst, p, bytes);
                p += bytes / sizeof(float);
                break;
        }
        off += align256(bytes);
    });
    BOA_REQUIRE((size_t)(p - w) == expect, "internal: weight cursor mismatch");
    if (net->wsets.size() >= 8) {  // bounded cache: drop the oldest set
        hipFree(net->wsets.front().arena);
        net->wsets.erase(net->wsets.begin());
    }
    unsigned char* arena = nullptr;
    BOA_TRY(boa_malloc_raw(c, total, (void**)&arena));
    hipError_t e = hipMemcpy(arena, stage.data(), total, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        hipFree(arena);
        BOA_HIP_TRY(e);
    }
    net->wsets.push_back({w, n_floats, hsh, arena, scales});
    point_layers_at(net, arena, scales);
    return BOA_OK;
}

extern "C" void boa_net_destroy(boa_net* net) {
    if (!net) return;
    hipStreamSynchronize(net->ctx->stream);
    for (void* a : net->allocs) hipFree(a);
    for (auto& ws : net->wsets) hipFree(ws.arena);
    delete net;
}

extern "C" int boa_net_create(boa_ctx* ctx, const boa_net_desc* desc, const float* host_weights, size_t n_floats,
                              int max_batch, int precision, boa_net** out) {
    BOA_REQUIRE(ctx && desc && out, "boa_net_create: NULL argument");
    BOA_REQUIRE(desc_ok(desc), "boa_net_create: invalid network geometry");
    BOA_REQUIRE(precision >= 0 && precision <= 2,
                "boa_net_create: precision %d not supported (0 = f16 MFMA / fp32 accumulate, 1 = fp32 reference mode, 2 = split-precision fp32)",
                precision);
    BOA_REQUIRE(max_batch >= 1 && max_batch <= 64, "boa_net_create: max_batch %d out of range", max_batch);
    BOA_HIP_TRY(hipSetDevice(ctx->device));
    boa_net* net = new boa_net();
    net->ctx = ctx;
    net->d = *desc;
    if (net->d.norm_eps <= 0.f) net->d.norm_eps = 1e-5f;
    if (net->d.lrelu_slope == 0.f) net->d.lrelu_slope = 0.01f;
    net->maxN = max_batch;
    net->precision = precision;
    const boa_net_desc& d = net->d;
    int rc = BOA_OK;
    auto fail = [&](int r) {
        boa_net_destroy(net);
        return r;
    };
    // encoder
    int din[3] = {d.patch[0], d.patch[1], d.patch[2]};
    int cin = d.in_channels;
    net->enc.resize(d.n_stages);
    for (int s = 0; s < d.n_stages; ++s) {
        net->enc[s].resize(d.n_conv_enc[s]);
        for (int i = 0; i < d.n_conv_enc[s]; ++i) {
            int one[3] = {1, 1, 1};
            const int* st = (i == 0) ? d.stride[s] : one;
            bool first = (s == 0 && i == 0);
            rc = setup_conv(net, net->enc[s][i], max_batch, din, cin, 0, d.features[s], d.kernel[s], st, first);
            if (rc) return fail(rc);
            const ConvGeom& g = net->enc[s][i].g;
            din[0] = g.Do; din[1] = g.Ho; din[2] = g.Wo;
            cin = d.features[s];
        }
        net->dims[s][0] = din[0]; net->dims[s][1] = din[1]; net->dims[s][2] = din[2];
    }
    // decoder
    net->up.resize(d.n_stages - 1);
    net->dec.resize(d.n_stages - 1);
    for (int k = 0; k < d.n_stages - 1; ++k) {
        int sb = d.n_stages - 1 - k;
        UpLayer& U = net->up[k];
        U.Cin = d.features[sb];
        U.Cout = d.features[sb - 1];
        for (int a = 0; a < 3; ++a) {
            U.s[a] = d.stride[sb][a];
            U.din[a] = net->dims[sb][a];
        }
        int dup[3] = {U.din[0] * U.s[0], U.din[1] * U.s[1], U.din[2] * U.s[2]};
        for (int a = 0; a < 3; ++a)
            if (dup[a] != net->dims[sb - 1][a]) {
                boa_set_error("decoder %d: upsampled dim %d (%d) != skip dim (%d); patch not divisible by strides", k, a,
                              dup[a], net->dims[sb - 1][a]);
                return fail(BOA_EINVAL);
            }
        if ((precision == 0 && U.Cin % 16) || (precision == 2 && U.Cin % 8) || U.Cout % 32) {
            boa_set_error("transposed conv %d -> %d: unsupported channel counts", U.Cin, U.Cout);
            return fail(BOA_EINVAL);
        }
        size_t vox = (size_t)dup[0] * dup[1] * dup[2];
        if (precision >= 1) {
            if ((rc = net_alloc_act(net, (size_t)max_batch * vox * U.Cout * sizeof(float), (void**)&U.out32))) return fail(rc);
        } else if ((rc = net_alloc_act(net, (size_t)max_batch * vox * U.Cout * sizeof(__half), (void**)&U.out)))
            return fail(rc);
        net->dec[k].resize(d.n_conv_dec[k]);
        int one[3] = {1, 1, 1};
        for (int i = 0; i < d.n_conv_dec[k]; ++i) {
            int c0 = U.Cout, c1 = (i == 0) ? U.Cout : 0;
            rc = setup_conv(net, net->dec[k][i], max_batch, dup, c0, c1, U.Cout, d.kernel[sb - 1], one, false);
            if (rc) return fail(rc);
        }
    }
    if ((rc = net_alloc(net, (size_t)max_batch * 3 * sizeof(int), (void**)&net->dev_origins))) return fail(rc);
    if (precision == 1) {
        if ((rc = net_alloc_act(net, (size_t)max_batch * d.in_channels * d.patch[0] * d.patch[1] * d.patch[2] * sizeof(float),
                                (void**)&net->tiles32)))
            return fail(rc);
    } else {
        int PD[3];
        conv_first_padded_dims(d.patch, d.kernel[0], PD);
        if ((rc = net_alloc_act(net, (size_t)max_batch * d.in_channels * PD[0] * PD[1] * PD[2] * sizeof(float),
                                (void**)&net->first_padded)))
            return fail(rc);
    }
    if ((rc = net_bind_arena(net))) return fail(rc);
    if (host_weights) {
        rc = boa_net_load_weights(net, host_weights, n_floats);
        if (rc) return fail(rc);
    }
    *out = net;
    return BOA_OK;
}

// fp32 mode: the same layer sequence through net_f32.hip; leaves the last decoder activation in dec.back().back().out32
static int net_forward_stack_f32(boa_net* net, const float* volume, const int V[3], const int vol_off[3],
                                 const int* host_origins, int N, int flip_mask) {
    boa_ctx* c = net->ctx;
    const boa_net_desc& d = net->d;
    BOA_REQUIRE(N >= 1 && N <= net->maxN, "forward: batch %d exceeds max_batch %d", N, net->maxN);
    BOA_HIP_TRY(hipMemcpyAsync(net->dev_origins, host_origins, (size_t)N * 3 * sizeof(int), hipMemcpyHostToDevice, c->stream));
    c->prof_break = true;
    BOA_TRY(launch_gather_tiles_f32(c, volume, V, vol_off, net->dev_origins, N, d.in_channels, d.patch, net->tiles32, flip_mask));
    struct Src {
        const float* data = nullptr;
        const float* ss = nullptr;
        int C = 0;
    };
    auto run_conv = [&](ConvLayer& L, const Src& a, const Src& b) -> int {
        const int din[3] = {L.g.Di, L.g.Hi, L.g.Wi}, dout[3] = {L.g.Do, L.g.Ho, L.g.Wo};
        BOA_TRY(launch_conv_f32(c, a.data, a.ss, a.C, b.data, b.ss, b.C, N, din, dout, L.g.k, L.g.s, L.g.Cout, L.w32, L.bias,
                                d.lrelu_slope, L.out32));
        return launch_stats_f32(c, L.out32, N, (size_t)dout[0] * dout[1] * dout[2], L.g.Cout, L.gamma, L.beta, d.norm_eps, L.ss);
    };
    Src cur, none;
    cur.data = net->tiles32;
    cur.C = d.in_channels;
    for (int s = 0; s < d.n_stages; ++s)
        for (size_t i = 0; i < net->enc[s].size(); ++i) {
            ConvLayer& L = net->enc[s][i];
            BOA_TRY(run_conv(L, cur, none));
            cur.data = L.out32; cur.ss = L.ss; cur.C = L.g.Cout;
        }
    for (int k = 0; k < d.n_stages - 1; ++k) {
        int sb = d.n_stages - 1 - k;
        UpLayer& U = net->up[k];
        BOA_TRY(launch_convt_f32(c, cur.data, cur.ss, U.Cin, N, U.din, U.s, U.Cout, U.w32, U.bias, d.lrelu_slope, U.out32));
        ConvLayer& SK = net->enc[sb - 1].back();
        Src upsrc, skip;
        upsrc.data = U.out32; upsrc.C = U.Cout;
        skip.data = SK.out32; skip.ss = SK.ss; skip.C = SK.g.Cout;
        for (size_t i = 0; i < net->dec[k].size(); ++i) {
            ConvLayer& L = net->dec[k][i];
            if (i == 0)
                BOA_TRY(run_conv(L, upsrc, skip));
            else
                BOA_TRY(run_conv(L, cur, none));
            cur.data = L.out32; cur.ss = L.ss; cur.C = L.g.Cout;
        }
    }
    return BOA_OK;
}

// split-precision mode: first conv on the fp32 VALU (k_conv_first<F32OUT>), 3x3x3 convs on k_conv_ws<X3>, transposed convs on
// k_convt_x3; InstanceNorm statistics from the conv epilogues (fp32 partial sums, fp64 finalize).  Leaves the last decoder
// activation (octet planes) in dec.back().back().out32.
static int net_forward_stack_x3(boa_net* net, const float* volume, const int V[3], const int vol_off[3], const int* host_origins,
                                int N, int flip_mask) {
    boa_ctx* c = net->ctx;
    const boa_net_desc& d = net->d;
    BOA_REQUIRE(N >= 1 && N <= net->maxN, "forward: batch %d exceeds max_batch %d", N, net->maxN);
    BOA_HIP_TRY(hipMemcpyAsync(net->dev_origins, host_origins, (size_t)N * 3 * sizeof(int), hipMemcpyHostToDevice, c->stream));
    c->prof_break = true;
    static const bool layer_prof = getenv("BOA_LAYER_PROF") != nullptr;
    auto prof_begin = [&]() {
        if (layer_prof) hipEventRecord(c->t0[7], c->stream);
    };
    auto prof_end = [&](const char* what, const int* din, int cin, int cout, const int* k, const int* s, double flops, const ConvTile* t) {
        if (!layer_prof) return;
        hipEventRecord(c->t1[7], c->stream);
        hipEventSynchronize(c->t1[7]);
        float ms = 0.f;
        hipEventElapsedTime(&ms, c->t0[7], c->t1[7]);
        fprintf(stderr, "[layer] x3 %-6s N=%d in=%dx%dx%d cin=%d cout=%d k=%d%d%d s=%d%d%d ", what, N, din[0], din[1], din[2], cin, cout, k[0], k[1],
                k[2], s[0], s[1], s[2]);
        if (t)
            fprintf(stderr, "R=%d w=%d,%d,%d b=%d,%d,%d tiles=%d lds=%zu ", t->R, t->w[0], t->w[1], t->w[2], t->b[0], t->b[1], t->b[2],
                    t->tiles[0] * t->tiles[1] * t->tiles[2], t->lds_bytes);
        fprintf(stderr, "%.1f us %.1f TFLOP/s\n", ms * 1e3, flops / (ms * 1e-3) / 1e12);
    };
    struct Src {
        const float* data = nullptr;
        const float* ss = nullptr;
        int C = 0;
    };
    auto run_conv = [&](ConvLayer& L, const Src& a, const Src& b) -> int {
        ConvGeom g = L.g;
        g.N = N;
        prof_begin();
        if (L.first) {
            int nblk = 0;
            BOA_TRY(launch_conv_first(c, volume, V, vol_off, net->dev_origins, N, d.in_channels, d.patch, L.g.k, L.g.Cout, L.wfirst, L.bias,
                                      net->first_padded, nullptr, L.partials, &nblk, flip_mask, L.out32));
        } else {
            BOA_TRY(launch_conv_x3(c, a.data, a.ss, a.C, b.data, b.ss, b.C, g, L.t, L.wpk, L.wscale, L.bias, d.lrelu_slope, L.out32, L.partials));
        }
        {
            const int din[3] = {g.Di, g.Hi, g.Wi};
            const double fl = 2.0 * N * (double)g.Do * g.Ho * g.Wo * g.k[0] * g.k[1] * g.k[2] * (L.Cin0 + L.Cin1) * g.Cout;
            prof_end(L.first ? "first" : "conv", din, L.Cin0 + L.Cin1, g.Cout, g.k, g.s, fl, L.first ? nullptr : &L.t);
        }
        return launch_norm_finalize(c, L.partials, L.nblk, N, g.Cout, (double)g.Do * g.Ho * g.Wo, L.gamma, L.beta, d.norm_eps, L.ss, nullptr, 1);
    };
    Src cur, none;
    for (int s = 0; s < d.n_stages; ++s)
        for (size_t i = 0; i < net->enc[s].size(); ++i) {
            ConvLayer& L = net->enc[s][i];
            BOA_TRY(run_conv(L, cur, none));
            cur.data = L.out32; cur.ss = L.ss; cur.C = L.g.Cout;
        }
    for (int k = 0; k < d.n_stages - 1; ++k) {
        int sb = d.n_stages - 1 - k;
        UpLayer& U = net->up[k];
        prof_begin();
        BOA_TRY(launch_convt_x3(c, cur.data, cur.ss, U.Cin, N, U.din, U.s, U.Cout, U.wpk, U.wscale, U.bias, d.lrelu_slope, U.out32));
        prof_end("convT", U.din, U.Cin, U.Cout, U.s, U.s, 2.0 * N * (double)U.din[0] * U.din[1] * U.din[2] * U.s[0] * U.s[1] * U.s[2] * U.Cin * U.Cout,
                 nullptr);
        ConvLayer& SK = net->enc[sb - 1].back();
        Src upsrc, skip;
        upsrc.data = U.out32; upsrc.C = U.Cout;
        skip.data = SK.out32; skip.ss = SK.ss; skip.C = SK.g.Cout;
        for (size_t i = 0; i < net->dec[k].size(); ++i) {
            ConvLayer& L = net->dec[k][i];
            if (i == 0)
                BOA_TRY(run_conv(L, upsrc, skip));
            else
                BOA_TRY(run_conv(L, cur, none));
            cur.data = L.out32; cur.ss = L.ss; cur.C = L.g.Cout;
        }
    }
    return BOA_OK;
}

// head of tile i of the current batch (either precision)
static int net_head(boa_net* net, int i, const int P[3], int plane_skip, float* logits_out, const uint16_t* gauss, uint16_t* acc,
                    uint16_t* nacc, const int PV[3], const int start[3]) {
    const boa_net_desc& d = net->d;
    ConvLayer& last = net->dec.back().back();
    const size_t pv = (size_t)d.patch[0] * d.patch[1] * d.patch[2];
    const float* ss = last.ss + (size_t)i * d.features[0] * 2;
    if (net->precision == 1)   // fp32 mode: channels-last records
        return launch_head_f32(net->ctx, last.out32 + (size_t)i * pv * d.features[0] + (size_t)plane_skip * d.patch[1] * d.patch[2] * d.features[0],
                               ss, d.features[0], P, d.num_classes, net->head_w, net->head_b, d.lrelu_slope, logits_out, gauss, acc,
                               nacc, PV, start);
    if (net->precision == 2) {  // split-precision mode: fp32 octet planes; skipped axis-0 planes are an offset inside every plane
        const float* a32 = last.out32 + (size_t)i * pv * d.features[0] + (size_t)plane_skip * d.patch[1] * d.patch[2] * 8;
        if (d.features[0] == 32 && d.num_classes <= 32)   // the gather head's arithmetic (label path == logits API, bit for bit)
            return launch_head_x3(net->ctx, a32, ss, d.features[0], P, d.num_classes, net->head_w, net->head_b, d.lrelu_slope, logits_out, gauss, acc,
                                  nacc, PV, start, pv);
        return launch_head_f32(net->ctx, a32, ss, d.features[0], P, d.num_classes, net->head_w, net->head_b, d.lrelu_slope, logits_out, gauss, acc, nacc,
                               PV, start, pv);
    }
    // chunk-planar fp16: skipping leading axis-0 planes is an offset inside every 16-channel plane; plane stride = whole tile
    return launch_head(net->ctx, last.out + (size_t)i * pv * d.features[0] + (size_t)plane_skip * d.patch[1] * d.patch[2] * 16, ss,
                       d.features[0], P, d.num_classes, net->head_w, net->head_b, d.lrelu_slope, logits_out, gauss, acc, nacc, PV,
                       start, pv);
}

// run the conv stack for N tiles; leaves the last decoder activation (+ its ss) in net->dec.back().back()
static int net_forward_stack(boa_net* net, const float* volume, const int V[3], const int vol_off[3],
                             const int* host_origins, int N, int flip_mask = 0) {
    if (net->precision == 1) return net_forward_stack_f32(net, volume, V, vol_off, host_origins, N, flip_mask);
    if (net->precision == 2) return net_forward_stack_x3(net, volume, V, vol_off, host_origins, N, flip_mask);
    boa_ctx* c = net->ctx;
    const boa_net_desc& d = net->d;
    BOA_REQUIRE(N >= 1 && N <= net->maxN, "forward: batch %d exceeds max_batch %d", N, net->maxN);
    BOA_HIP_TRY(hipMemcpyAsync(net->dev_origins, host_origins, (size_t)N * 3 * sizeof(int), hipMemcpyHostToDevice,
                               c->stream));
    static const bool layer_prof = getenv("BOA_LAYER_PROF") != nullptr;
    auto prof_begin = [&]() {
        if (layer_prof) hipEventRecord(c->t0[7], c->stream);
    };
    auto prof_end = [&](const char* what, const int* din, int cin, int cout, const int* k, const int* s, double flops,
                        const ConvTile* t) {
        if (!layer_prof) return;
        hipEventRecord(c->t1[7], c->stream);
        hipEventSynchronize(c->t1[7]);
        float ms = 0.f;
        hipEventElapsedTime(&ms, c->t0[7], c->t1[7]);
        fprintf(stderr, "[layer] %-6s N=%d in=%dx%dx%d cin=%d cout=%d k=%d%d%d s=%d%d%d ", what, N, din[0], din[1], din[2], cin,
                cout, k[0], k[1], k[2], s[0], s[1], s[2]);
        if (t)
            fprintf(stderr, "var=%d R=%d w=%d,%d,%d b=%d,%d,%d tiles=%d lds=%zu ", t->variant, t->R, t->w[0], t->w[1], t->w[2],
                    t->b[0], t->b[1], t->b[2], t->tiles[0] * t->tiles[1] * t->tiles[2], t->lds_bytes);
        fprintf(stderr, "%.1f us %.1f TFLOP/s\n", ms * 1e3, flops / (ms * 1e-3) / 1e12);
    };
    auto run_conv = [&](ConvLayer& L, const ActSrc& a, const ActSrc& b) -> int {
        ConvGeom g = L.g;
        g.N = N;
        prof_begin();
        if (L.first) {
            int nblk = 0;
            BOA_TRY(launch_conv_first(c, volume, V, vol_off, net->dev_origins, N, d.in_channels, d.patch, L.g.k, L.g.Cout,
                                      L.wfirst, L.bias, net->first_padded, L.out, L.partials, &nblk, flip_mask));
        } else {
            BOA_TRY(launch_conv_mfma(c, a, b, g, L.t, L.wpk, L.bias, d.lrelu_slope, L.out, L.partials));
            // BOA_LAYER_PROF_REPEAT=n [BOA_LAYER_PROF_MATCH=Di,Cin,Cout]: the same launch n more times, timed as one block (sustained
            // clocks; tools/power_sample.sh samples the socket power meanwhile)
            static const int prof_repeat = getenv("BOA_LAYER_PROF_REPEAT") ? atoi(getenv("BOA_LAYER_PROF_REPEAT")) : 0;
            static int m_di = -1, m_ci = -1, m_co = -1;
            static const bool has_match = getenv("BOA_LAYER_PROF_MATCH") && sscanf(getenv("BOA_LAYER_PROF_MATCH"), "%d,%d,%d", &m_di, &m_ci, &m_co) == 3;
            if (layer_prof && prof_repeat > 0 && (!has_match || (g.Di == m_di && L.Cin0 + L.Cin1 == m_ci && g.Cout == m_co))) {
                prof_begin();
                for (int rep = 0; rep < prof_repeat; ++rep)
                    BOA_TRY(launch_conv_mfma(c, a, b, g, L.t, L.wpk, L.bias, d.lrelu_slope, L.out, L.partials));
                hipEventRecord(c->t1[7], c->stream);
                hipEventSynchronize(c->t1[7]);
                float ms = 0.f;
                hipEventElapsedTime(&ms, c->t0[7], c->t1[7]);
                fprintf(stderr, "[repeat] in=%d cin=%d cout=%d: %d launches, %.1f us each, %.3f s\n", g.Di, L.Cin0 + L.Cin1, g.Cout, prof_repeat,
                        ms * 1e3 / prof_repeat, ms * 1e-3);
                prof_begin();
                BOA_TRY(launch_conv_mfma(c, a, b, g, L.t, L.wpk, L.bias, d.lrelu_slope, L.out, L.partials));
            }
        }
        {
            const int din[3] = {g.Di, g.Hi, g.Wi};
            const double fl = 2.0 * N * (double)g.Do * g.Ho * g.Wo * g.k[0] * g.k[1] * g.k[2] * (L.Cin0 + L.Cin1) * g.Cout;
            prof_end(L.first ? "first" : "conv", din, L.Cin0 + L.Cin1, g.Cout, g.k, g.s, fl, L.first ? nullptr : &L.t);
        }
        double count = (double)g.Do * g.Ho * g.Wo;
        BOA_TRY(launch_norm_finalize(c, L.partials, L.nblk, N, g.Cout, count, L.gamma, L.beta, d.norm_eps, L.ss, L.ss16, 1));
        return BOA_OK;
    };
    ActSrc cur, none;
    for (int s = 0; s < d.n_stages; ++s)
        for (size_t i = 0; i < net->enc[s].size(); ++i) {
            ConvLayer& L = net->enc[s][i];
            BOA_TRY(run_conv(L, cur, none));
            cur.data = L.out;
            cur.ss = L.ss;
            cur.ss16 = L.ss16;
            cur.C = L.g.Cout;
        }
    for (int k = 0; k < d.n_stages - 1; ++k) {
        int sb = d.n_stages - 1 - k;
        UpLayer& U = net->up[k];
        prof_begin();
        BOA_TRY(launch_convt_mfma(c, cur, N, U.din, U.s, U.Cout, U.wpk, U.bias, d.lrelu_slope, U.out));
        prof_end("convT", U.din, U.Cin, U.Cout, U.s, U.s,
                 2.0 * N * (double)U.din[0] * U.din[1] * U.din[2] * U.s[0] * U.s[1] * U.s[2] * U.Cin * U.Cout, nullptr);
        ConvLayer& SK = net->enc[sb - 1].back();
        ActSrc upsrc, skip;
        upsrc.data = U.out; upsrc.ss = nullptr; upsrc.ss16 = nullptr; upsrc.C = U.Cout;
        skip.data = SK.out; skip.ss = SK.ss; skip.ss16 = SK.ss16; skip.C = SK.g.Cout;
        for (size_t i = 0; i < net->dec[k].size(); ++i) {
            ConvLayer& L = net->dec[k][i];
            if (i == 0)
                BOA_TRY(run_conv(L, upsrc, skip));
            else
                BOA_TRY(run_conv(L, cur, none));
            cur.data = L.out;
            cur.ss = L.ss;
            cur.ss16 = L.ss16;
            cur.C = L.g.Cout;
        }
    }
    return BOA_OK;
}

// `_internal_maybe_mirror_and_predict` (predict_from_raw_data.py:541-557) for the nb tiles of one batch: the fp32 logits of
// the plain forward plus, for every non-empty combination of the allowed mirror axes (itertools.combinations order: single
// axes, pairs, the triple), the logits of the flipped tile flipped back, divided by the number of variants.  Result in
// net->mirror_sum [nb][C][P].
static int net_mirrored_logits(boa_net* net, const float* volume, const int V[3], const int vol_off[3], const int* host_origins, int nb) {
    const boa_net_desc& d = net->d;
    const size_t pv = (size_t)d.patch[0] * d.patch[1] * d.patch[2], per = (size_t)d.num_classes * pv;
    if (!net->mirror_tmp) {
        BOA_TRY(net_alloc(net, (size_t)net->maxN * per * sizeof(float), (void**)&net->mirror_tmp));
        BOA_TRY(net_alloc(net, (size_t)net->maxN * per * sizeof(float), (void**)&net->mirror_sum));
    }
    std::vector<int> combos = {0};
    int axes[3], na = 0;
    for (int a = 0; a < 3; ++a)
        if (net->mirror_mask & (1 << a)) axes[na++] = a;
    for (int size = 1; size <= na; ++size)
        for (int m = 1; m < (1 << na); ++m) {   // subsets of `size` axes in lexicographic order of their axis tuples
            if (__builtin_popcount(m) != size) continue;
            combos.push_back(m);
        }
    // lexicographic order of tuples: for size 1: (a0), (a1), (a2); size 2: (a0,a1), (a0,a2), (a1,a2) = ascending bit masks
    // 3, 5, 6 -- the ascending-mask enumeration above already yields that order for up to three axes
    for (size_t k = 0; k < combos.size(); ++k) {
        int flip = 0;
        for (int j = 0; j < na; ++j)
            if (combos[k] & (1 << j)) flip |= 1 << axes[j];
        BOA_TRY(net_forward_stack(net, volume, V, vol_off, host_origins, nb, flip));
        const bool last = k + 1 == combos.size();
        for (int i = 0; i < nb; ++i) {
            BOA_TRY(net_head(net, i, d.patch, 0, net->mirror_tmp + (size_t)i * per, nullptr, nullptr, nullptr, nullptr, nullptr));
            BOA_TRY(launch_flip_accumulate(net->ctx, net->mirror_tmp + (size_t)i * per, net->mirror_sum + (size_t)i * per, d.num_classes,
                                           d.patch, flip, k > 0, last ? (float)combos.size() : 1.0f));
        }
    }
    return BOA_OK;
}

extern "C" int boa_net_set_mirroring(boa_net* net, int axes_mask) {
    BOA_REQUIRE(net && axes_mask >= 0 && axes_mask < 8, "boa_net_set_mirroring: axes mask %d", axes_mask);
    net->mirror_mask = axes_mask;
    return BOA_OK;
}

extern "C" int boa_net_forward(boa_net* net, const float* dev_volume, const int V[3], const int* host_origins,
                               int n_tiles, float* dev_logits_out) {
    BOA_REQUIRE(net && dev_volume && V && host_origins && dev_logits_out, "boa_net_forward: NULL argument");
    BOA_TRY(net_bind_arena(net));
    const boa_net_desc& d = net->d;
    const size_t pv = (size_t)d.patch[0] * d.patch[1] * d.patch[2];
    const int zero[3] = {0, 0, 0};
    for (int t0 = 0; t0 < n_tiles; t0 += net->maxN) {
        int nb = std::min(net->maxN, n_tiles - t0);
        if (net->mirror_mask) {
            BOA_TRY(net_mirrored_logits(net, dev_volume, V, zero, host_origins + (size_t)t0 * 3, nb));
            BOA_HIP_TRY(hipMemcpyAsync(dev_logits_out + (size_t)t0 * d.num_classes * pv, net->mirror_sum,
                                       (size_t)nb * d.num_classes * pv * sizeof(float), hipMemcpyDeviceToDevice, net->ctx->stream));
            net->ctx->prof_break = true;
            continue;
        }
        BOA_TRY(net_forward_stack(net, dev_volume, V, zero, host_origins + (size_t)t0 * 3, nb));
        for (int i = 0; i < nb; ++i)
            BOA_TRY(net_head(net, i, d.patch, 0, dev_logits_out + (size_t)(t0 + i) * d.num_classes * pv, nullptr, nullptr, nullptr,
                             nullptr, nullptr));
    }
    return BOA_OK;
}

extern "C" int boa_net_predict_sliding_window(boa_net* net, const float* dev_volume, const int V[3], const int PV[3],
                                              const int* vol_off, const int* host_origins, int n_tiles,
                                              const uint16_t* dev_gauss, uint16_t* dev_acc, uint16_t* dev_n) {
    BOA_REQUIRE(net && dev_volume && V && PV && host_origins && dev_acc && dev_n,
                "boa_net_predict_sliding_window: NULL argument");
    BOA_TRY(net_bind_arena(net));
    const boa_net_desc& d = net->d;
    const int zero[3] = {0, 0, 0};
    const int* off = vol_off ? vol_off : zero;
    for (int a = 0; a < 3; ++a)
        BOA_REQUIRE(PV[a] >= d.patch[a] && off[a] >= 0 && off[a] + V[a] <= PV[a],
                    "sliding window: padded dim %d (%d) must cover patch (%d) and volume (%d at %d)", a, PV[a],
                    d.patch[a], V[a], off[a]);
    const size_t pv = (size_t)d.patch[0] * d.patch[1] * d.patch[2];
    for (int t0 = 0; t0 < n_tiles; t0 += net->maxN) {
        int nb = std::min(net->maxN, n_tiles - t0);
        if (net->mirror_mask) {  // mirrored mean of the fp32 logits first, then the reference's accumulate step on it
            BOA_TRY(net_mirrored_logits(net, dev_volume, V, off, host_origins + (size_t)t0 * 3, nb));
            for (int i = 0; i < nb; ++i)
                BOA_TRY(boa_accumulate_tile(net->ctx, net->mirror_sum + (size_t)i * d.num_classes * pv, dev_gauss, dev_acc, dev_n,
                                            d.num_classes, d.patch, PV, host_origins + (size_t)(t0 + i) * 3));
            continue;
        }
        BOA_TRY(net_forward_stack(net, dev_volume, V, off, host_origins + (size_t)t0 * 3, nb));
        for (int i = 0; i < nb; ++i) {  // canonical order: one launch per tile, serialised on the stream
            const int* st = host_origins + (size_t)(t0 + i) * 3;
            BOA_TRY(net_head(net, i, d.patch, 0, nullptr, dev_gauss, dev_acc, dev_n, PV, st));
        }
    }
    return BOA_OK;
}

// ------------------------------------------------------------------------------------------------------
// Fused sliding window -> labels (head_gather.hip): the conv stack writes the last decoder activation of EVERY tile of the
// volume into the context's stash, then one gather pass per fold walks the volume.  Conditions (else the caller uses
// boa_net_predict_sliding_window + boa_finalize_labels): production or split-precision mode, no test-time mirroring, features[0] == 32,
// <= 32 classes, tile origins = the full cartesian grid of per-axis steps in canonical (x outer, z inner) order.
static bool grid_origins(const int* o, int n, std::vector<int> (&steps)[3]) {
    for (int a = 0; a < 3; ++a) steps[a].clear();
    if (n < 1) return false;
    // canonical order: the last axis varies fastest
    for (int t = 0; t < n && (t == 0 || o[t * 3 + 2] > o[(t - 1) * 3 + 2]); ++t) steps[2].push_back(o[t * 3 + 2]);
    const int n2 = (int)steps[2].size();
    if (n % n2) return false;
    for (int t = 0; t < n; t += n2) {
        if (t > 0 && o[t * 3 + 1] <= o[(t - n2) * 3 + 1]) break;
        steps[1].push_back(o[t * 3 + 1]);
    }
    const int n1 = (int)steps[1].size();
    if (n % (n1 * n2)) return false;
    for (int t = 0; t < n; t += n1 * n2) steps[0].push_back(o[t * 3]);
    const int n0 = (int)steps[0].size();
    if ((long long)n0 * n1 * n2 != n) return false;
    for (int a = 0; a < 3; ++a)
        for (size_t i = 1; i < steps[a].size(); ++i)
            if (steps[a][i] <= steps[a][i - 1]) return false;
    for (int t = 0; t < n; ++t) {
        const int iz = t % n2, iy = (t / n2) % n1, ix = t / (n1 * n2);
        if (o[t * 3] != steps[0][ix] || o[t * 3 + 1] != steps[1][iy] || o[t * 3 + 2] != steps[2][iz]) return false;
    }
    return n0 < 256 && n1 < 256 && n2 < 256;
}

// The last decoder activation of EVERY tile of a fold, kept in the context's stash for the gather head (k_gather_head): layout
// [activations][fp32 ss][packed ss16 of the conv stack][head ss table][walk table]
struct TileStash {
    const __half* act = nullptr;   // fp16 chunk planes (fp32 octet planes in the split-precision mode)
    float* ss = nullptr;           // [tile][F0][2] fp32 (scale, shift) of the last InstanceNorm
    unsigned* ssp = nullptr;       // [tile][2][16] packed fp16 (scale, shift): the fp16 head's table
    int* tab = nullptr;            // walk table (device)
    bool x3 = false;
    std::vector<int> steps[3];     // tile origins per axis
};


// Keeps the context's tile stash out of boa_trim's reach (an allocation that fails under memory pressure trims, and a trim frees an
// idle stash) from before the tiles are written until the LAST consumer of the TileStash pointers -- the deferred planes' copies, the
// gather head launch -- is queued on the stream; a later trim synchronises the stream before it frees anything.
struct StashHold {
    boa_ctx* c;
    explicit StashHold(boa_ctx* ctx) : c(ctx) { c->stash_busy = true; }
    ~StashHold() { c->stash_busy = false; }
    StashHold(const StashHold&) = delete;
    StashHold& operator=(const StashHold&) = delete;
};

// runs the conv stack over all tiles (batches of max_batch), the last decoder conv writing straight into the stash slots of its tiles, and
// builds the gather head's walk table.  BOA_ENOMEM when the stash does not fit (the caller falls back to the scatter form).
static int net_forward_into_stash(boa_net* net, const float* dev_volume, const int V[3], const int PV[3], const int* off, const int* host_origins,
                                  int n_tiles, TileStash& ts) {
    boa_ctx* c = net->ctx;
    const boa_net_desc& d = net->d;
    std::vector<int> (&steps)[3] = ts.steps;
    grid_origins(host_origins, n_tiles, steps);
    const int F0 = d.features[0];
    const size_t pv = (size_t)d.patch[0] * d.patch[1] * d.patch[2];
    // stash layout: [activations][fp32 ss][packed ss16 of the conv stack][head ss table][steps]
    auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const bool x3 = net->precision == 2;   // split-precision mode: the stash holds the fp32 octet planes, the head reads the fp32 (scale, shift)
    const size_t o_act = 0, o_ss = align((size_t)n_tiles * pv * F0 * (x3 ? sizeof(float) : sizeof(__half))), o_ss16 = align(o_ss + (size_t)n_tiles * F0 * 2 * sizeof(float)),
                 o_ssp = align(o_ss16 + (size_t)n_tiles * F0 * sizeof(unsigned)), o_steps = align(o_ssp + (size_t)n_tiles * 32 * sizeof(unsigned)),
                 need = align(o_steps + ((size_t)n_tiles + PV[0] + PV[1] + PV[2] + 64) * sizeof(int));
    if (c->stash_bytes < need) {
        BOA_HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->stash) hipFree(c->stash);
        c->stash = nullptr;
        c->stash_bytes = 0;
        boa_trim(c);
        // The stash may take a bounded share of what is free NOW ($BOA_STASH_FRAC, default 0.6): what follows the network on this
        // context and on the GPU's other contexts -- fold buffers, post-processing volumes, the second lane, RCCL buffers -- has no
        // fallback of its own, the tile loop has one (BOA_ENOMEM here sends the caller to the scatter form, which needs
        // (C + 1) fp16 planes instead of a tile stash).
        {
            static const double frac = getenv("BOA_STASH_FRAC") ? atof(getenv("BOA_STASH_FRAC")) : 0.6;
            size_t fr = 0, tot = 0;
            if (hipMemGetInfo(&fr, &tot) == hipSuccess && (double)need > frac * (double)fr) {
                boa_set_error("fused sliding window: stash of %zu bytes exceeds %.2f of the %zu free bytes", need, frac, fr);
                return BOA_ENOMEM;
            }
        }
        if (hipMalloc(&c->stash, need) != hipSuccess) {
            (void)hipGetLastError();
            c->stash = nullptr;
            boa_set_error("fused sliding window: %zu bytes of stash do not fit", need);
            return BOA_ENOMEM;
        }
        c->stash_bytes = need;
    }
    unsigned char* base = (unsigned char*)c->stash;
    __half* s_act = (__half*)(base + o_act);
    ts.x3 = x3;
    float* s_ss = (float*)(base + o_ss);
    unsigned* s_ss16 = (unsigned*)(base + o_ss16);
    unsigned* s_ssp = (unsigned*)(base + o_ssp);
    int* s_steps = (int*)(base + o_steps);
    ConvLayer& last = net->dec.back().back();
    __half* keep_out = last.out;
    float* keep_out32 = last.out32;
    float* keep_ss = last.ss;
    unsigned* keep_ss16 = last.ss16;
    int rc = BOA_OK;
    // (the caller holds c->stash_busy -- StashHold -- until every use of the returned pointers has been queued: boa_trim must not
    //  release the stash while tiles are being written into it, nor between this call and the caller's copies / gather launch)
    for (int t0 = 0; t0 < n_tiles && rc == BOA_OK; t0 += net->maxN) {
        const int nb = std::min(net->maxN, n_tiles - t0);
        // the last decoder conv of this batch writes straight into the stash slots of its tiles
        if (x3)
            last.out32 = (float*)s_act + (size_t)t0 * pv * F0;
        else
            last.out = s_act + (size_t)t0 * pv * F0;
        last.ss = s_ss + (size_t)t0 * F0 * 2;
        last.ss16 = s_ss16 + (size_t)t0 * F0;
        rc = net_forward_stack(net, dev_volume, V, off, host_origins + (size_t)t0 * 3, nb);
    }
    last.out = keep_out;
    last.out32 = keep_out32;
    last.ss = keep_ss;
    last.ss16 = keep_ss16;
    if (rc) return rc;
    if (!x3) BOA_TRY(launch_pack_head_ss(c, s_ss, s_ssp, n_tiles));
    // walk table: tile origins per axis, then per coordinate the first covering tile and the count (x, y), per 32-voxel z run the
    // tiles that intersect the run
    std::vector<int> tab;
    for (int a = 0; a < 3; ++a)
        for (int v : steps[a]) tab.push_back(v);
    auto cover = [&](int a, int lo, int hi) {
        int first = 0, cnt = 0;
        for (size_t i = 0; i < steps[a].size(); ++i)
            if (steps[a][i] <= hi && steps[a][i] + d.patch[a] > lo) {
                if (!cnt) first = (int)i;
                ++cnt;
            }
        return first | (cnt << 8);
    };
    for (int x = 0; x < PV[0]; ++x) tab.push_back(cover(0, x, x));
    for (int y = 0; y < PV[1]; ++y) tab.push_back(cover(1, y, y));
    for (int zb = 0; zb < PV[2]; zb += 32) tab.push_back(cover(2, zb, std::min(zb + 31, PV[2] - 1)));
    BOA_HIP_TRY(hipMemcpyAsync(s_steps, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    BOA_HIP_TRY(hipStreamSynchronize(c->stream));   // (the table is a stack-lifetime host vector)
    c->prof_break = true;
    ts.act = s_act;
    ts.ss = s_ss;
    ts.ssp = s_ssp;
    ts.tab = s_steps;
    return BOA_OK;
}

extern "C" int boa_net_labels_supported(boa_net* net, const int* host_origins, int n_tiles) {
    if (!net || !host_origins) return 0;
    static const bool off = getenv("BOA_NO_GATHER_HEAD") != nullptr;
    // (patch z extent a multiple of 32 and <= 31 classes: the shapes for which the scatter loop's head runs on the matrix cores too,
    //  so that the label path and the logits API share one head arithmetic)
    if (off || net->precision == 1 || net->mirror_mask != 0 || net->d.features[0] != 32 || net->d.num_classes > 31 || net->d.patch[2] % 32 != 0) return 0;
    std::vector<int> steps[3];
    return grid_origins(host_origins, n_tiles, steps) ? 1 : 0;
}

extern "C" int boa_net_predict_labels_fold(boa_net* net, const float* dev_volume, const int V[3], const int PV[3], const int* vol_off,
                                           const int* host_origins, int n_tiles, const uint16_t* dev_gauss, uint16_t* dev_fold,
                                           int fold_index, int n_folds, const uint8_t* host_lut, int merge, uint8_t* dev_labels_out,
                                           const int* crop_off, const int* crop_dims, int* dev_inf_flag) {
    BOA_REQUIRE(net && dev_volume && V && PV && host_origins && dev_inf_flag, "boa_net_predict_labels_fold: NULL argument");
    BOA_REQUIRE(boa_net_labels_supported(net, host_origins, n_tiles), "boa_net_predict_labels_fold: unsupported network / tile layout");
    BOA_TRY(net_bind_arena(net));
    BOA_REQUIRE(n_folds >= 1 && fold_index >= 0 && fold_index < n_folds && (n_folds == 1 || dev_fold), "boa_net_predict_labels_fold: folds");
    BOA_REQUIRE(fold_index + 1 < n_folds || dev_labels_out, "boa_net_predict_labels_fold: the last fold needs the label buffer");
    boa_ctx* c = net->ctx;
    const boa_net_desc& d = net->d;
    const int zero[3] = {0, 0, 0};
    const int* off = vol_off ? vol_off : zero;
    for (int a = 0; a < 3; ++a)
        BOA_REQUIRE(PV[a] >= d.patch[a] && off[a] >= 0 && off[a] + V[a] <= PV[a],
                    "fused sliding window: padded dim %d (%d) must cover patch (%d) and volume (%d at %d)", a, PV[a], d.patch[a], V[a], off[a]);
    TileStash ts;
    StashHold hold(net->ctx);   // until launch_gather_head below is queued
    BOA_TRY(net_forward_into_stash(net, dev_volume, V, PV, off, host_origins, n_tiles, ts));
    float* s_ss = ts.ss;
    unsigned* s_ssp = ts.ssp;
    int* s_steps = ts.tab;
    const __half* s_act = ts.act;
    const bool x3 = ts.x3;
    std::vector<int> (&steps)[3] = ts.steps;
    const int ntile[3] = {(int)steps[0].size(), (int)steps[1].size(), (int)steps[2].size()};
    const int mode = n_folds == 1 ? 0 : (fold_index == 0 ? 1 : (fold_index + 1 == n_folds ? 3 : 2));
    return launch_gather_head(c, s_act, x3 ? (const unsigned*)s_ss : s_ssp, net->head_w, net->head_b, dev_gauss, d.num_classes, d.patch, PV, ntile,
                              s_steps, dev_fold, mode, n_folds, host_lut, merge, dev_labels_out, crop_off, crop_dims, dev_inf_flag, d.lrelu_slope,
                              n_tiles, x3);
}

// ------------------------------------------------------------------------------------------------------
// tile-sharded sliding window (several GPUs on one volume): the rank that owns tile rows [b0, b1) along
// axis 0 cannot add the first `defer` planes of its row-b0 tiles before the lower rank's partial sums for those
// planes have arrived (the reference's fp16 `+=` runs in ascending tile order per voxel).  The head input of those
// planes is kept in a stash and applied afterwards; everything else is accumulated at once.
struct boa_stash {
    boa_ctx* ctx = nullptr;
    unsigned char* arena = nullptr;
    // head weights of the weight set that produced the stashed activations: the stash may be applied after the network has
    // switched to the next fold's weights (the exchange of fold f overlaps the tiles of fold f + 1); the sets are cached device
    // arenas (boa_net::wsets), so the pointers outlive the switch
    const float* head_w = nullptr;
    const float* head_b = nullptr;
    struct Item {
        size_t act_off, ss_off;
        int planes;
        int start[3];
    };
    std::vector<Item> items;
    // gather form (boa_net_predict_sliding_window_deferred ran the gather head): the first dp planes of every deferring tile (the block's
    // first tile row; with steps below half a patch also the rows behind it, which defer fewer planes) in the gather head's own stash
    // layout -- [tile][F / 16 planes][dp * P1 * P2 voxels][32 B], the (scale, shift) tables, the walk table -- so that
    // boa_net_apply_deferred is ONE more k_gather_head launch over planes [x0, x_split), started from the lower rank's sums
    bool gather = false, x3 = false;
    int dp = 0, x0 = 0, x_split = 0, n0 = 0, n1 = 0, n2 = 0, n_items = 0;
    size_t o_ss = 0, o_ssp = 0, o_tab = 0;
};

extern "C" void boa_stash_destroy(boa_stash* st) {
    if (!st) return;
    if (st->arena) boa_free(st->ctx, st->arena);
    delete st;
}

extern "C" int boa_net_predict_sliding_window_deferred(boa_net* net, const float* dev_volume, const int V[3],
                                                       const int PV[3], const int* vol_off, const int* host_origins,
                                                       int n_tiles, const uint16_t* dev_gauss, uint16_t* dev_acc,
                                                       uint16_t* dev_n, const int* host_defer_planes,
                                                       boa_stash** stash_out) {
    BOA_REQUIRE(net && dev_volume && V && PV && host_origins && dev_acc && dev_n && host_defer_planes && stash_out,
                "boa_net_predict_sliding_window_deferred: NULL argument");
    BOA_TRY(net_bind_arena(net));
    const boa_net_desc& d = net->d;
    BOA_REQUIRE(net->mirror_mask == 0, "deferred sliding window (tile sharding) is not available with test-time mirroring");
    const int zero[3] = {0, 0, 0};
    const int* off = vol_off ? vol_off : zero;
    for (int a = 0; a < 3; ++a)
        BOA_REQUIRE(PV[a] >= d.patch[a] && off[a] >= 0 && off[a] + V[a] <= PV[a],
                    "sliding window: padded dim %d (%d) must cover patch (%d) and volume (%d at %d)", a, PV[a],
                    d.patch[a], V[a], off[a]);
    const int F = d.features[0];
    const size_t plane = (size_t)d.patch[1] * d.patch[2];
    const size_t pv = (size_t)d.patch[0] * plane;
    const bool f32 = net->precision == 1;   // fp32 reference mode: channels-last fp32 records, the first dp planes are a contiguous prefix
    const bool x3 = net->precision == 2;    // split-precision mode: fp32 octet planes (F / 8 planes of 32 bytes per voxel)
    const size_t esz = (f32 || x3) ? 4 : 2;
    boa_stash* st = new boa_stash;
    st->ctx = net->ctx;
    st->head_w = net->head_w;
    st->head_b = net->head_b;
    size_t bytes = 0;
    for (int i = 0; i < n_tiles; ++i) {
        int dp = host_defer_planes[i];
        if (dp < 0 || dp > d.patch[0]) {
            delete st;
            BOA_REQUIRE(false, "deferred sliding window: tile %d defers %d planes of %d", i, dp, d.patch[0]);
        }
        if (dp == 0) continue;
        boa_stash::Item it;
        it.act_off = bytes;
        bytes += ((size_t)dp * plane * F * esz + 255) / 256 * 256;
        it.ss_off = bytes;
        bytes += 256 * ((F * 2 * 4 + 255) / 256);
        it.planes = dp;
        for (int a = 0; a < 3; ++a) it.start[a] = host_origins[(size_t)i * 3 + a];
        st->items.push_back(it);
    }
    int rc = BOA_OK;
    // Gather form (the product path when the network / tile grid allow it, as in boa_net_predict_labels_fold): every tile's last
    // activation goes to the context's stash, the planes to defer are copied out of it, and ONE k_gather_head launch in raw mode
    // writes the partial sums of all other planes of this rank -- [x_split, end of its last row) -- instead of one accumulator
    // read-modify-write per covering tile.  The planes below x_split are exactly the deferred ones (checked) and all belong to the
    // block's first tile row: they stay untouched until boa_net_apply_deferred adds them, with the same kernel, on top of the
    // lower rank's sums.  Same head arithmetic for every tile (the matrix-core head), whatever the tile origins' alignment.
    static const bool shard_scatter = getenv("BOA_SHARD_SCATTER") != nullptr;   // experiment hook: the round-3 scatter loop
    if (!shard_scatter && !f32 && n_tiles > 0 && boa_net_labels_supported(net, host_origins, n_tiles)) {
        // dp0 = the deepest deferral (the block's first row); rows that start further up defer fewer planes -- actual steps below
        // half a patch make the block's second row reach the lower block's last row too.  Every deferred tile keeps dp0 planes (the
        // later rows more than they defer: valid planes of the tile, never visited by the launch over [x0, x_split)).
        int x_first = host_origins[0], x_split = -1, x_end = 0, dp0 = 0, x0 = 0, n_def = 0;
        std::vector<int> def_rows;
        bool consistent = true;
        for (int i = 0; i < n_tiles; ++i) {
            const int xo = host_origins[(size_t)i * 3], dpi = host_defer_planes[i];
            x_first = std::min(x_first, xo);
            x_end = std::max(x_end, xo + d.patch[0]);
            if (dpi > 0) {
                if (n_def == 0) {
                    dp0 = dpi;
                    x0 = xo;
                    x_split = xo + dpi;
                }
                consistent = consistent && xo + dpi == x_split && xo >= x0;   // all end at the same plane (canonical order: x0 first)
                if (def_rows.empty() || def_rows.back() != xo) def_rows.push_back(xo);
                ++n_def;
            }
        }
        if (x_split < 0) x_split = x_first;
        for (int i = 0; i < n_tiles; ++i) {   // every tile that reaches below x_split defers exactly its planes below x_split
            const int below = std::max(0, std::min(x_split - host_origins[(size_t)i * 3], d.patch[0]));
            consistent = consistent && host_defer_planes[i] == below;
        }
        TileStash ts;
        StashHold hold(net->ctx);   // across the arena allocation below (it may trim), the stash copies and the raw gather launch
        int grc = consistent ? net_forward_into_stash(net, dev_volume, V, PV, off, host_origins, n_tiles, ts) : BOA_ENOMEM;
        if (grc == BOA_OK) {
            boa_ctx* c = net->ctx;
            // the deferred planes in the gather head's layout
            st->gather = true;
            st->x3 = x3;
            st->dp = dp0;
            st->x0 = x0;
            st->n1 = (int)ts.steps[1].size();
            st->n2 = (int)ts.steps[2].size();
            st->n_items = n_def;
            auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
            const size_t item_act = (size_t)dp0 * plane * F * esz;
            st->o_ss = align((size_t)n_def * item_act);
            st->o_ssp = align(st->o_ss + (size_t)n_def * F * 2 * sizeof(float));
            st->o_tab = align(st->o_ssp + (size_t)n_def * 32 * sizeof(unsigned));
            st->n0 = (int)def_rows.size();
            st->x_split = x_split;
            const size_t gbytes = align(st->o_tab + (def_rows.size() + st->n1 + st->n2 + PV[0] + PV[1] + PV[2] / 32 + 8) * sizeof(int));
            if (n_def > 0) {
                if ((rc = boa_malloc(c, gbytes, (void**)&st->arena)) != BOA_OK) {
                    boa_stash_destroy(st);
                    return rc;
                }
                if (n_def != (int)def_rows.size() * st->n1 * st->n2) {
                    boa_stash_destroy(st);
                    boa_set_error("deferred sliding window: %d deferred tiles in %d rows of %d x %d", n_def, (int)def_rows.size(), st->n1, st->n2);
                    return BOA_EINVAL;
                }
                bool ok_copy = true;
                int item = 0;
                const int nplanes = x3 ? F / 8 : F / 16;   // 32-byte records per voxel and plane in both layouts
                for (int i = 0; i < n_tiles && ok_copy; ++i) {
                    if (host_defer_planes[i] == 0) continue;
                    const unsigned char* tile_act = (const unsigned char*)ts.act + (size_t)i * pv * F * esz;
                    for (int k = 0; k < nplanes && ok_copy; ++k)
                        ok_copy = hipMemcpyAsync(st->arena + (size_t)item * item_act + (size_t)k * dp0 * plane * 32, tile_act + (size_t)k * pv * 32,
                                                 (size_t)dp0 * plane * 32, hipMemcpyDeviceToDevice, c->stream) == hipSuccess;
                    ok_copy = ok_copy && hipMemcpyAsync(st->arena + st->o_ss + (size_t)item * F * 2 * sizeof(float), ts.ss + (size_t)i * F * 2,
                                                        (size_t)F * 2 * sizeof(float), hipMemcpyDeviceToDevice, c->stream) == hipSuccess;
                    ++item;
                }
                if (!ok_copy) {
                    boa_stash_destroy(st);
                    boa_set_error("deferred sliding window: stash copy failed");
                    return BOA_EHIP;
                }
                if (!x3 && (rc = launch_pack_head_ss(c, (const float*)(st->arena + st->o_ss), (unsigned*)(st->arena + st->o_ssp), n_def)) != BOA_OK) {
                    boa_stash_destroy(st);
                    return rc;
                }
                // walk table of the one-row tile grid with dp planes per tile
                std::vector<int> tab;
                for (int v : def_rows) tab.push_back(v);
                for (int v : ts.steps[1]) tab.push_back(v);
                for (int v : ts.steps[2]) tab.push_back(v);
                auto cover = [&](int a, int ext, int lo, int hi) {
                    const std::vector<int>& stp = a == 0 ? def_rows : ts.steps[a];
                    int first = 0, cnt = 0;
                    for (size_t k = 0; k < stp.size(); ++k)
                        if (stp[k] <= hi && stp[k] + ext > lo) {
                            if (!cnt) first = (int)k;
                            ++cnt;
                        }
                    return first | (cnt << 8);
                };
                for (int x = 0; x < PV[0]; ++x) tab.push_back(cover(0, dp0, x, x));
                for (int y = 0; y < PV[1]; ++y) tab.push_back(cover(1, d.patch[1], y, y));
                for (int zb = 0; zb < PV[2]; zb += 32) tab.push_back(cover(2, d.patch[2], zb, std::min(zb + 31, PV[2] - 1)));
                if (hipMemcpyAsync(st->arena + st->o_tab, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                    hipStreamSynchronize(c->stream) != hipSuccess) {   // (the table is a stack-lifetime host vector)
                    boa_stash_destroy(st);
                    boa_set_error("deferred sliding window: walk table copy failed");
                    return BOA_EHIP;
                }
            }
            c->prof_break = true;
            const int ntile[3] = {(int)ts.steps[0].size(), (int)ts.steps[1].size(), (int)ts.steps[2].size()};
            const int xr[2] = {x_split, std::min(x_end, PV[0])};
            rc = launch_gather_head(c, ts.act, x3 ? (const unsigned*)ts.ss : ts.ssp, net->head_w, net->head_b, dev_gauss, d.num_classes, d.patch, PV, ntile,
                                    ts.tab, dev_acc, 4, 1, nullptr, 0, nullptr, nullptr, nullptr, nullptr, d.lrelu_slope, n_tiles, x3, xr, dev_n, 0);
            if (rc != BOA_OK) {
                boa_stash_destroy(st);
                return rc;
            }
            *stash_out = st;
            return BOA_OK;
        }
        if (grc != BOA_ENOMEM) {
            boa_stash_destroy(st);
            return grc;
        }
        // (stash does not fit / unusual deferral pattern: the scatter loop below)
    }
    rc = bytes ? boa_malloc(net->ctx, bytes, (void**)&st->arena) : BOA_OK;
    if (rc != BOA_OK) {
        delete st;
        return rc;
    }
    size_t item = 0;
    for (int t0 = 0; t0 < n_tiles && rc == BOA_OK; t0 += net->maxN) {
        int nb = std::min(net->maxN, n_tiles - t0);
        rc = net_forward_stack(net, dev_volume, V, off, host_origins + (size_t)t0 * 3, nb);
        ConvLayer& last = net->dec.back().back();
        for (int i = 0; i < nb && rc == BOA_OK; ++i) {
            const int* stt = host_origins + (size_t)(t0 + i) * 3;
            const __half* act = (f32 || x3) ? nullptr : last.out + (size_t)i * pv * F;
            const float* ss = last.ss + (size_t)i * F * 2;
            int dp = host_defer_planes[t0 + i];
            if (dp > 0) {
                const boa_stash::Item& it = st->items[item++];
                // chunk-planar: the first dp axis-0 planes of every 16-channel plane; the stash keeps them planar with its own
                // plane stride (dp * plane voxels)
                bool ok_copy = true;
                if (f32)
                    ok_copy = hipMemcpyAsync(st->arena + it.act_off, last.out32 + (size_t)i * pv * F, (size_t)dp * plane * F * 4,
                                             hipMemcpyDeviceToDevice, net->ctx->stream) == hipSuccess;
                for (int k = 0; x3 && k < F / 8 && ok_copy; ++k)
                    ok_copy = hipMemcpyAsync(st->arena + it.act_off + (size_t)k * dp * plane * 32, last.out32 + (size_t)i * pv * F + (size_t)k * pv * 8,
                                             (size_t)dp * plane * 32, hipMemcpyDeviceToDevice, net->ctx->stream) == hipSuccess;
                for (int k = 0; !f32 && !x3 && k < F / 16 && ok_copy; ++k)
                    ok_copy = hipMemcpyAsync(st->arena + it.act_off + (size_t)k * dp * plane * 32, act + (size_t)k * pv * 16,
                                             (size_t)dp * plane * 32, hipMemcpyDeviceToDevice, net->ctx->stream) == hipSuccess;
                if (!ok_copy ||
                    hipMemcpyAsync(st->arena + it.ss_off, ss, (size_t)F * 2 * 4, hipMemcpyDeviceToDevice,
                                   net->ctx->stream) != hipSuccess) {
                    boa_set_error("deferred sliding window: stash copy failed");
                    rc = BOA_EHIP;
                    break;
                }
                net->ctx->prof_break = true;
            }
            if (dp < d.patch[0]) {
                int P[3] = {d.patch[0] - dp, d.patch[1], d.patch[2]};
                int s2[3] = {stt[0] + dp, stt[1], stt[2]};
                rc = net_head(net, i, P, dp, nullptr, dev_gauss ? dev_gauss + (size_t)dp * plane : nullptr, dev_acc, dev_n, PV, s2);
            }
        }
    }
    if (rc != BOA_OK) {
        boa_stash_destroy(st);
        return rc;
    }
    *stash_out = st;
    return BOA_OK;
}

extern "C" int boa_net_apply_deferred(boa_net* net, const boa_stash* st, const uint16_t* dev_gauss, uint16_t* dev_acc,
                                      uint16_t* dev_n, const int PV[3]) {
    BOA_REQUIRE(net && st && dev_acc && dev_n && PV, "boa_net_apply_deferred: NULL argument");
    const boa_net_desc& d = net->d;
    if (st->gather) {
        if (st->n_items == 0) return BOA_OK;
        const int P[3] = {st->dp, d.patch[1], d.patch[2]};
        const int ntile[3] = {st->n0, st->n1, st->n2};
        const int xr[2] = {st->x0, std::min(st->x_split, PV[0])};
        return launch_gather_head(net->ctx, (const __half*)st->arena, (const unsigned*)(st->arena + (st->x3 ? st->o_ss : st->o_ssp)), st->head_w, st->head_b,
                                  dev_gauss, d.num_classes, P, PV, ntile, (const int*)(st->arena + st->o_tab), dev_acc, 4, 1, nullptr, 0, nullptr, nullptr,
                                  nullptr, nullptr, d.lrelu_slope, st->n_items, st->x3, xr, dev_n, 1);
    }
    for (const boa_stash::Item& it : st->items) {  // the stash keeps the canonical tile order
        int P[3] = {it.planes, d.patch[1], d.patch[2]};
        if (net->precision == 1)
            BOA_TRY(launch_head_f32(net->ctx, (const float*)(st->arena + it.act_off), (const float*)(st->arena + it.ss_off),
                                    d.features[0], P, d.num_classes, st->head_w, st->head_b, d.lrelu_slope, nullptr, dev_gauss,
                                    dev_acc, dev_n, PV, it.start));
        else if (net->precision == 2 && d.features[0] == 32 && d.num_classes <= 32)
            BOA_TRY(launch_head_x3(net->ctx, (const float*)(st->arena + it.act_off), (const float*)(st->arena + it.ss_off),
                                   d.features[0], P, d.num_classes, st->head_w, st->head_b, d.lrelu_slope, nullptr, dev_gauss,
                                   dev_acc, dev_n, PV, it.start, (size_t)it.planes * d.patch[1] * d.patch[2]));
        else if (net->precision == 2)
            BOA_TRY(launch_head_f32(net->ctx, (const float*)(st->arena + it.act_off), (const float*)(st->arena + it.ss_off),
                                    d.features[0], P, d.num_classes, st->head_w, st->head_b, d.lrelu_slope, nullptr, dev_gauss,
                                    dev_acc, dev_n, PV, it.start, (size_t)it.planes * d.patch[1] * d.patch[2]));
        else
            BOA_TRY(launch_head(net->ctx, (const __half*)(st->arena + it.act_off), (const float*)(st->arena + it.ss_off),
                                d.features[0], P, d.num_classes, st->head_w, st->head_b, d.lrelu_slope, nullptr, dev_gauss,
                                dev_acc, dev_n, PV, it.start, (size_t)it.planes * d.patch[1] * d.patch[2]));
    }
    return BOA_OK;
}

// ------------------------------------------------------------------------------------------------------
// unit-test seams
extern "C" int boa_conv_block_test(boa_ctx* ctx, const float* dev_in, int N, int Cin, const int dims[3],
                                   const float* host_w, const float* host_b, const float* host_gamma,
                                   const float* host_beta, int Cout, const int kernel[3], const int stride[3],
                                   int with_norm_act, int impl, float* dev_out) {
    BOA_REQUIRE(ctx && dev_in && dims && host_w && host_b && kernel && stride && dev_out, "conv test: NULL argument");
    BOA_REQUIRE(impl == 0, "conv test: impl %d not available", impl);
    ConvGeom g;
    g.N = N; g.Di = dims[0]; g.Hi = dims[1]; g.Wi = dims[2]; g.Cout = Cout; g.Cin = Cin;
    int dout[3];
    for (int a = 0; a < 3; ++a) {
        g.k[a] = kernel[a];
        g.s[a] = stride[a];
        dout[a] = (dims[a] + 2 * ((kernel[a] - 1) / 2) - kernel[a]) / stride[a] + 1;
    }
    g.Do = dout[0]; g.Ho = dout[1]; g.Wo = dout[2];
    ConvTile t;
    ConvGeom gref = g;
    gref.N = tile_ref_batch();  // as the network does: the tile shape must not depend on the batch size
    BOA_REQUIRE(choose_conv_tile(gref, ctx->cu_count, &t), "conv test: no tile configuration");
    size_t vin = (size_t)dims[0] * dims[1] * dims[2], vout = (size_t)dout[0] * dout[1] * dout[2];
    __half *in16 = nullptr, *out16 = nullptr, *wpk = nullptr;
    float *bias = nullptr, *gamma = nullptr, *beta = nullptr, *partials = nullptr, *ss = nullptr;
    int nblk = conv_nblk(t, ctx->cu_count, Cout);
    std::vector<__half> tmp(conv_wpk_halves(Cin, Cout, kernel));
    pack_conv_weights(host_w, Cin, Cout, kernel, tmp.data());
    std::vector<float> ones(Cout, 1.f), zeros(Cout, 0.f);
    int rc = BOA_OK;
#define T_(x) do { if (rc == BOA_OK) rc = (x); } while (0)
    T_(boa_malloc(ctx, (size_t)N * vin * Cin * 2, (void**)&in16));
    T_(boa_malloc(ctx, (size_t)N * vout * Cout * 2, (void**)&out16));
    T_(boa_malloc(ctx, tmp.size() * 2, (void**)&wpk));
    T_(boa_malloc(ctx, Cout * 4, (void**)&bias));
    T_(boa_malloc(ctx, Cout * 4, (void**)&gamma));
    T_(boa_malloc(ctx, Cout * 4, (void**)&beta));
    T_(boa_malloc(ctx, (size_t)N * Cout * 2 * nblk * 4, (void**)&partials));
    T_(boa_memset(ctx, partials, 0, (size_t)N * Cout * 2 * nblk * 4));
    T_(boa_malloc(ctx, (size_t)N * Cout * 2 * 4, (void**)&ss));
    T_(boa_h2d(ctx, wpk, tmp.data(), tmp.size() * 2));
    T_(boa_h2d(ctx, bias, host_b, Cout * 4));
    T_(boa_h2d(ctx, gamma, host_gamma ? host_gamma : ones.data(), Cout * 4));
    T_(boa_h2d(ctx, beta, host_beta ? host_beta : zeros.data(), Cout * 4));
    T_(launch_nchw_to_ndhwc_f16(ctx, dev_in, N, Cin, vin, in16));
    ActSrc a, none;
    a.data = in16; a.ss = nullptr; a.C = Cin;
    T_(launch_conv_mfma(ctx, a, none, g, t, wpk, bias, 0.01f, out16, partials));
    T_(launch_norm_finalize(ctx, partials, nblk, N, Cout, (double)vout, gamma, beta, 1e-5f, ss, nullptr, 1));
    T_(launch_ndhwc_to_nchw_f32(ctx, out16, with_norm_act ? ss : nullptr, 0.01f, N, Cout, vout, dev_out));
    if (rc == BOA_OK) rc = boa_sync(ctx);
#undef T_
    boa_free(ctx, in16); boa_free(ctx, out16); boa_free(ctx, wpk); boa_free(ctx, bias); boa_free(ctx, gamma);
    boa_free(ctx, beta); boa_free(ctx, partials); boa_free(ctx, ss);
    return rc;
}

extern "C" int boa_net_debug_activation(boa_net* net, int kind, int stage, int conv, int tile, float* dev_out, int* channels_out,
                                        int dims_out[3]) {
    BOA_REQUIRE(net && channels_out && dims_out, "boa_net_debug_activation: NULL argument");
    BOA_REQUIRE(tile >= 0 && tile < net->maxN, "boa_net_debug_activation: tile %d outside the batch", tile);
    BOA_TRY(net_bind_arena(net));
    const float* ss = nullptr;
    const __half* a16 = nullptr;
    const float* a32 = nullptr;
    int Cc = 0, dm[3] = {0, 0, 0};
    if (kind == 1) {
        BOA_REQUIRE(stage >= 0 && stage < (int)net->up.size(), "boa_net_debug_activation: no transposed conv %d", stage);
        const UpLayer& U = net->up[stage];
        Cc = U.Cout;
        for (int a = 0; a < 3; ++a) dm[a] = U.din[a] * U.s[a];
        a16 = U.out;
        a32 = U.out32;
    } else {
        auto& stages = kind == 0 ? net->enc : net->dec;
        BOA_REQUIRE((kind == 0 || kind == 2) && stage >= 0 && stage < (int)stages.size() && conv >= 0 && conv < (int)stages[stage].size(),
                    "boa_net_debug_activation: no layer (kind %d, stage %d, conv %d)", kind, stage, conv);
        const ConvLayer& L = stages[stage][conv];
        Cc = L.g.Cout;
        dm[0] = L.g.Do; dm[1] = L.g.Ho; dm[2] = L.g.Wo;
        a16 = L.out;
        a32 = L.out32;
        ss = L.ss + (size_t)tile * Cc * 2;
    }
    *channels_out = Cc;
    for (int a = 0; a < 3; ++a) dims_out[a] = dm[a];
    if (!dev_out) return BOA_OK;  // size query
    const size_t vox = (size_t)dm[0] * dm[1] * dm[2];
    if (net->precision == 1)
        return launch_ndhwc32_to_nchw_f32(net->ctx, a32 + (size_t)tile * vox * Cc, ss, net->d.lrelu_slope, Cc, vox, dev_out);
    if (net->precision == 2)
        return launch_octet_to_nchw_f32(net->ctx, a32 + (size_t)tile * vox * Cc, ss, net->d.lrelu_slope, Cc, vox, dev_out);
    return launch_ndhwc_to_nchw_f32(net->ctx, a16 + (size_t)tile * vox * Cc, ss, net->d.lrelu_slope, 1, Cc, vox, dev_out);
}

extern "C" int boa_head_tile(boa_ctx* ctx, const uint16_t* dev_act, const float* dev_ss, int F0, const int P[3], int C,
                             const float* dev_w, const float* dev_b, float slope, float* dev_logits_out,
                             const uint16_t* dev_gauss, uint16_t* dev_acc, uint16_t* dev_n, const int PV[3],
                             const int start[3]) {
    BOA_REQUIRE(ctx && dev_act && dev_ss && P && dev_w && dev_b, "boa_head_tile: NULL argument");
    BOA_REQUIRE(dev_logits_out || (dev_acc && dev_n && PV && start), "boa_head_tile: neither logits_out nor accumulators given");
    return launch_head(ctx, (const __half*)dev_act, dev_ss, F0, P, C, dev_w, dev_b, slope, dev_logits_out, dev_gauss, dev_acc,
                       dev_n, PV, start);
}

extern "C" int boa_convtranspose_test(boa_ctx* ctx, const float* dev_in, int N, int Cin, const int dims[3],
                                      const float* host_w, const float* host_b, int Cout, const int stride[3],
                                      float* dev_out) {
    BOA_REQUIRE(ctx && dev_in && dims && host_w && host_b && stride && dev_out, "convT test: NULL argument");
    size_t vin = (size_t)dims[0] * dims[1] * dims[2];
    size_t vout = vin * stride[0] * stride[1] * stride[2];
    __half *in16 = nullptr, *out16 = nullptr, *wpk = nullptr;
    float* bias = nullptr;
    std::vector<__half> tmp(convt_wpk_halves(Cin, Cout, stride));
    pack_convt_weights(host_w, Cin, Cout, stride, tmp.data());
    int rc = BOA_OK;
#define T_(x) do { if (rc == BOA_OK) rc = (x); } while (0)
    T_(boa_malloc(ctx, (size_t)N * vin * Cin * 2, (void**)&in16));
    T_(boa_malloc(ctx, (size_t)N * vout * Cout * 2, (void**)&out16));
    T_(boa_malloc(ctx, tmp.size() * 2, (void**)&wpk));
    T_(boa_malloc(ctx, Cout * 4, (void**)&bias));
    T_(boa_h2d(ctx, wpk, tmp.data(), tmp.size() * 2));
    T_(boa_h2d(ctx, bias, host_b, Cout * 4));
    T_(launch_nchw_to_ndhwc_f16(ctx, dev_in, N, Cin, vin, in16));
    ActSrc a;
    a.data = in16; a.ss = nullptr; a.C = Cin;
    T_(launch_convt_mfma(ctx, a, N, dims, stride, Cout, wpk, bias, 0.01f, out16));
    T_(launch_ndhwc_to_nchw_f32(ctx, out16, nullptr, 0.01f, N, Cout, vout, dev_out));
    if (rc == BOA_OK) rc = boa_sync(ctx);
#undef T_
    boa_free(ctx, in16); boa_free(ctx, out16); boa_free(ctx, wpk); boa_free(ctx, bias);
    return rc;
}
