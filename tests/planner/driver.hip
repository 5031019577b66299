// Driver of the HOST-ONLY sanitizer build (tests/planner/build.py: every source of the library compiled with
// --offload-host-only -DGSCAN_PLAN_ONLY -fsanitize=address,undefined).  It calls the C ABI the way the Python host does, for
// every shape of a list, with real host allocations standing in for device memory: the library's sequencing and planning
// code (workspace_layout, check_dims, pick_split, GemmBatch::add / launch / launch_macro_tiles, the decoders' LDS budgets,
// every argument check) runs instrumented; device calls are no-ops and kernel launches are checked against the hardware
// limits and counted (common.h, GSCAN_PLAN_ONLY).  Nothing here ever touches a GPU.
//
//   planner_asan <shapes.txt>      one shape per line:  H He E k Co cond aux bi layers B G L T Vi V C
// Exit code 0: every call returned (0, or 1 with a message from the library: a rejected shape is a valid outcome);
// 3: a launch broke a hardware limit.  A sanitizer finding aborts the process with its own report.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "step.h"

namespace gscan {
static long g_launches = 0, g_bad_launches = 0;
void plan_record(const char *kernel, dim3 grid, dim3 block, size_t lds_bytes) {
    ++g_launches;
    const unsigned long long threads = (unsigned long long)block.x * block.y * block.z;
    const unsigned long long groups = (unsigned long long)grid.x * grid.y * grid.z;
    if (groups == 0 || groups >= (1ull << 31) || threads == 0 || threads > 1024 || lds_bytes > 160 * 1024) {
        ++g_bad_launches;
        fprintf(stderr, "BAD LAUNCH %s: grid %u x %u x %u, block %u x %u x %u, %zu bytes of LDS\n", kernel, grid.x, grid.y,
                grid.z, block.x, block.y, block.z, lds_bytes);
    }
}
}  // namespace gscan

namespace {
struct Arena {          // stands in for device memory: allocated, never touched by the library's host code
    std::vector<void *> blocks;
    template <typename T> T *get(size_t n) {
        void *p = nullptr;
        if (posix_memalign(&p, 256, (n ? n : 1) * sizeof(T)) != 0) { fprintf(stderr, "out of memory\n"); exit(2); }
        blocks.push_back(p);
        return (T *)p;
    }
    ~Arena() { for (void *p : blocks) free(p); }
};

// the flat parameter buffer of the Python host: named_parameters() order, every parameter on a 16-byte boundary
struct Flat {
    float *base; size_t at = 0;
    float *take(size_t n) { at = (at + 3) / 4 * 4; float *p = base ? base + at : nullptr; at += n; return p; }   // null base: sizing pass
};
size_t fill(gscan_params &p, const gscan_dims &d, float *base) {
    Flat f{base};
    const size_t C = d.C, Co = d.Co, k = d.K3, E = d.E, He = d.He, H = d.H, F = 3 * Co, D = d.bidirectional ? 2 : 1;
    memset(&p, 0, sizeof(p));
    p.conv1_w = f.take(Co * C); p.conv1_b = f.take(Co); p.conv2_w = f.take(Co * C * 25); p.conv2_b = f.take(Co);
    p.conv3_w = f.take(Co * C * k * k); p.conv3_b = f.take(Co);
    p.vis_key_w = f.take(H * F); p.vis_query_w = f.take(H * H); p.vis_energy_w = f.take(H);
    p.enc_emb = f.take((size_t)d.Vi * E);
    p.enc_w_ih = f.take(4 * He * E); p.enc_w_hh = f.take(4 * He * He); p.enc_b_ih = f.take(4 * He); p.enc_b_hh = f.take(4 * He);
    if (D == 2) { p.enc_w_ih_rev = f.take(4 * He * E); p.enc_w_hh_rev = f.take(4 * He * He); p.enc_b_ih_rev = f.take(4 * He); p.enc_b_hh_rev = f.take(4 * He); }
    for (int l = 1; l < (d.enc_layers > 1 ? d.enc_layers : 1); ++l)
        for (int dir = 0; dir < (int)D; ++dir) {
            p.enc_deep[l - 1][4 * dir + 0] = f.take(4 * He * D * He); p.enc_deep[l - 1][4 * dir + 1] = f.take(4 * He * He);
            p.enc_deep[l - 1][4 * dir + 2] = f.take(4 * He); p.enc_deep[l - 1][4 * dir + 3] = f.take(4 * He);
        }
    p.bridge_w = f.take(H * He); p.bridge_b = f.take(H);
    p.txt_key_w = f.take(H * He); p.txt_query_w = f.take(H * H); p.txt_energy_w = f.take(H);
    if (d.conditional) { p.q2k_w = f.take(H * 2 * H); p.q2k_b = f.take(H); }
    p.dec_emb = f.take((size_t)d.V * H);
    p.dec_w_ih = f.take(4 * H * 3 * H); p.dec_w_hh = f.take(4 * H * H); p.dec_b_ih = f.take(4 * H); p.dec_b_hh = f.take(4 * H);
    p.out2hid_w = f.take(H * 4 * H); p.hid2out_w = f.take((size_t)d.V * H);
    return f.at;
}

long g_calls = 0, g_rejected = 0;
void report(const char *what, int rc, const gscan_dims &d) {
    ++g_calls;
    if (rc != 0) {
        ++g_rejected;
        printf("  rejected %s (H=%d He=%d B=%d G=%d L=%d T=%d): %s\n", what, d.H, d.He, d.B, d.G, d.L, d.T, gscan_last_error());
    }
}

void one_shape(gscan_dims d) {
    Arena mem;
    const size_t bytes = gscan_workspace_bytes(&d);
    if (bytes == 0) { report("workspace_bytes", 1, d); return; }
    size_t off = 0, cnt = 0;
    report("workspace_find", gscan_workspace_find(&d, "S", &off, &cnt), d);
    const size_t B = d.B, L = d.L, T = d.T, M = (size_t)d.G * d.G, V = d.V, H = d.H, He = d.He, F = 3 * (size_t)d.Co,
                 D = d.bidirectional ? 2 : 1, layers = d.enc_layers > 1 ? d.enc_layers : 1;
    float *ws = mem.get<float>(bytes / 4);
    gscan_params p, g;
    const size_t np = fill(p, d, nullptr);                     // size first (offsets from a null base are never used)
    float *pbuf = mem.get<float>(np + 8), *gbuf = mem.get<float>(np + 8);
    fill(p, d, pbuf);
    fill(g, d, gbuf);
    gscan_batch bt{};
    bt.commands = mem.get<int64_t>(B * L); bt.cmd_lengths = mem.get<int32_t>(B); bt.targets = mem.get<int64_t>(B * T);
    bt.target_positions = mem.get<int64_t>(B);
    bt.world = mem.get<float>(B * M * d.C);
    gscan_masks mk{mem.get<float>(B * M * F), mem.get<float>(B * L * d.E), mem.get<float>(B * T * H),
                   layers > 1 ? mem.get<float>((layers - 1) * B * L * D * He) : nullptr};
    float *logp = mem.get<float>(B * T * V), *aux = mem.get<float>(B * M), *stats = mem.get<float>(4), *seeds = mem.get<float>(3);
    report("train_step_nll(mean)", gscan_train_step_nll(&d, &p, &bt, &mk, ws, logp, aux, 0.3f, 0, stats, seeds, &g, nullptr), d);
    report("train_step_nll(sum)", gscan_train_step_nll(&d, &p, &bt, nullptr, ws, logp, aux, 0.3f, 1, stats, seeds, &g, nullptr), d);
    report("forward", gscan_forward(&d, &p, &bt, &mk, ws, logp, aux, nullptr), d);
    report("backward", gscan_backward(&d, &p, &bt, &mk, ws, logp, d.auxiliary ? aux : nullptr, &g, nullptr), d);
    report("backward_seeded", gscan_backward_seeded(&d, &p, &bt, &mk, ws, logp, d.auxiliary ? aux : nullptr, seeds, &g, nullptr), d);
    report("backward_nll", gscan_backward_nll(&d, &p, &bt, &mk, ws, 0.3f, 0, stats, seeds, &g, nullptr), d);
    {   // the uint8 world of the batcher
        gscan_batch b8 = bt;
        b8.world = nullptr; b8.world_u8 = mem.get<uint8_t>(B * M * d.C);
        report("forward(u8 world)", gscan_forward(&d, &p, &b8, nullptr, ws, logp, aux, nullptr), d);
    }
    report("decode_batched", gscan_decode_batched(&d, &p, &bt, mem.get<float>(B * M * F), mem.get<float>(B * L * He),
                                                  mem.get<float>(B * He), ws, logp, aux, nullptr), d);
    report("step_losses", gscan_step_losses(logp, bt.targets, d.auxiliary ? aux : nullptr, bt.target_positions, d.B, d.T, d.V,
                                            (int)M, d.pad_tgt, stats, mem.get<float>(B * T * V), mem.get<float>(B * M), nullptr), d);
    report("sequence_metrics", gscan_sequence_metrics(logp, bt.targets, d.B, d.T, d.V, d.pad_tgt, seeds, nullptr), d);
    report("adam_step_masks", gscan_adam_step_masks(pbuf, gbuf, mem.get<float>(np), mem.get<float>(np), np, 1e-3f, 0.9f, 0.999f, 1e-8f,
                                                    0.9f, 20000.f, 3, nullptr, mem.get<float>(B * M * F + B * L * d.E + B * T * H),
                                                    B * M * F, B * L * d.E, B * T * H, 0.1f, 0.3f, 0.3f, 1234, 7, nullptr), d);
    // greedy decoding: the T = 1 layout of the workspace
    gscan_dims d1 = d;
    d1.T = 1;
    const size_t bytes1 = gscan_workspace_bytes(&d1);
    if (bytes1) {
        float *ws1 = mem.get<float>(bytes1 / 4);
        const int max_steps = d.T + 1;
        report("encode", gscan_encode(&d1, &p, &bt, nullptr, ws1, nullptr), d1);
        report("decode_step", gscan_decode_step(&d1, &p, &bt, bt.targets, mem.get<float>(B * H), mem.get<float>(B * H), ws1,
                                                mem.get<float>(B * V), mem.get<float>(B * H), mem.get<float>(B * H),
                                                mem.get<float>(B * L), mem.get<float>(B * M), nullptr), d1);
        report("greedy_decode", gscan_greedy_decode(&d1, max_steps, &p, &bt, ws1, 1, 2 % d.V, mem.get<int64_t>(B * max_steps),
                                                    mem.get<int32_t>(B), mem.get<float>(B * max_steps * L),
                                                    mem.get<float>(B * max_steps * M), mem.get<float>(B * M), nullptr), d1);
    }
}

void gemm_shapes() {      // products of awkward extents through both GEMM entry points (split-K with and without slabs)
    Arena mem;
    const int dims[][4] = {{1, 1, 1, 1}, {33, 65, 31, 1}, {400, 300, 5120, 8}, {9, 400, 30720, 48}, {9216, 400, 150, 1},
                           {5120, 200, 500, 1}, {257, 17, 4099, 5}, {2048, 1040, 4096, 8}, {100, 100, 163, 2}};
    gscan_dims none{};
    for (auto &s : dims) {
        const int M = s[0], N = s[1], K = s[2], split = s[3];
        float *a = mem.get<float>((size_t)M * K), *b = mem.get<float>((size_t)K * N), *c = mem.get<float>((size_t)M * N);
        report("gemm nn", gscan_gemm_f32(M, N, K, 1.f, a, K, 1, b, N, 1, 0.f, c, N, nullptr, 0, nullptr, 1, nullptr), none);
        report("gemm tn split", gscan_gemm_f32(M, N, K, 0.5f, a, 1, M, b, N, 1, 1.f, c, N, nullptr, 0, nullptr, split, nullptr), none);
        float *scratch = mem.get<float>(16u << 20);
        report("gemm tn slabs", gscan_gemm_f32_scratch(M, N, K, 1.f, a, 1, M, b, N, 1, 1.f, c, N, nullptr, 0, nullptr, split,
                                                       mem.get<float>(M), scratch, 16u << 20, nullptr), none);
        report("gemm tn slabs (tiny scratch)", gscan_gemm_f32_scratch(M, N, K, 1.f, a, 1, M, b, N, 1, 1.f, c, N, nullptr, 0, nullptr,
                                                                     split, nullptr, scratch, 1000, nullptr), none);
    }
}
}  // namespace

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s shapes.txt\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "r");
    if (!f) { perror(argv[1]); return 2; }
    int H, He, E, k, Co, cond, aux, bi, layers, B, G, L, T, Vi, V, C, shapes = 0;
    while (fscanf(f, "%d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d", &H, &He, &E, &k, &Co, &cond, &aux, &bi, &layers, &B, &G,
                  &L, &T, &Vi, &V, &C) == 16) {
        gscan_dims d{};
        d.B = B; d.L = L; d.T = T; d.G = G; d.C = C; d.Co = Co; d.K3 = k; d.E = E; d.He = He; d.H = H; d.Vi = Vi; d.V = V;
        d.conditional = cond; d.auxiliary = aux; d.bidirectional = bi; d.pad_in = 0; d.pad_tgt = 0; d.enc_layers = layers;
        one_shape(d);
        ++shapes;
    }
    fclose(f);
    gemm_shapes();
    printf("shapes %d, library calls %ld (%ld rejected with a message), kernel launches planned %ld, bad launches %ld\n", shapes,
           g_calls, g_rejected, gscan::g_launches, gscan::g_bad_launches);
    return gscan::g_bad_launches ? 3 : 0;
}
