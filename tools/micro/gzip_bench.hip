// tools/micro/gzip_bench.hip -- the gzip kernels of rsq_deflate.h alone, on a text file: a few seconds to compile where the library takes minutes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I reseq_amd/csrc [-DNAME=...] -o /tmp/gzip_bench tools/micro/gzip_bench.hip -lz
//   /tmp/gzip_bench text.fq [copies [lines|dense]]
// The text, repeated `copies` times in device memory (default: up to 768 MB), through k_gzip_pieces<true> (sample) -> build_codes -> k_gzip_pieces<false> + k_gzip_stored:
// kernel time by HIP events, size against zlib levels 1 and 6, the first members inflated by zlib and compared with the text and with the host walk (piece_on_the_host).
#include <hip/hip_runtime.h>
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rsq_deflate.h"

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

using namespace rsq::gz;

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: gzip_bench text [copies]\n");
        return 2;
    }
    FILE *f = fopen(argv[1], "rb");
    if (!f) {
        perror(argv[1]);
        return 1;
    }
    std::vector<uint8_t> one;
    {
        uint8_t buf[1 << 16];
        size_t k;
        while ((k = fread(buf, 1, sizeof buf, f)) > 0) one.insert(one.end(), buf, buf + k);
        fclose(f);
    }
    const size_t copies = argc > 2 ? (size_t)atoll(argv[2]) : std::max<size_t>(1, ((size_t)768 << 20) / one.size());
    const uint64_t n = (uint64_t)one.size() * copies;
    uint8_t *text = nullptr;
    CHECK(hipMalloc(&text, n + 64));
    for (size_t c = 0; c < copies; ++c) CHECK(hipMemcpy(text + c * one.size(), one.data(), one.size(), hipMemcpyHostToDevice));
    const uint64_t n_pieces = (n + kPiece - 1) / kPiece;
    uint8_t *slots = nullptr;
    uint32_t *sizes = nullptr, *hist = nullptr;
    Codes *codes = nullptr;
    CHECK(hipMalloc(&slots, n_pieces * kSlot));
    CHECK(hipMalloc(&sizes, n_pieces * 4));
    CHECK(hipMalloc(&hist, (kLitLen + kDist) * 4));
    CHECK(hipMalloc(&codes, sizeof(Codes)));
    CHECK(hipMemset(hist, 0, (kLitLen + kDist) * 4));
    const uint32_t stride = sample_stride(n_pieces);
    uint32_t *hist2 = nullptr;
    CHECK(hipMalloc(&hist2, (kLitLen + kDist) * 4));
    CHECK(hipMemset(hist2, 0, (kLitLen + kDist) * 4));
    hipLaunchKernelGGL((k_gzip_pieces<true, kProbeStep>), dim3((n_pieces + stride - 1) / stride), dim3(kThreads), 0, 0, text, n, stride, (const Codes *)nullptr, (uint8_t *)nullptr, (uint32_t *)nullptr, hist);
    hipLaunchKernelGGL((k_gzip_pieces<true, kDenseStep>), dim3((n_pieces + stride - 1) / stride), dim3(kThreads), 0, 0, text, n, stride, (const Codes *)nullptr, (uint8_t *)nullptr, (uint32_t *)nullptr, hist2);
    std::vector<uint32_t> h_hist(kLitLen + kDist), h_hist2(kLitLen + kDist);
    CHECK(hipMemcpy(h_hist.data(), hist, h_hist.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(h_hist2.data(), hist2, h_hist2.size() * 4, hipMemcpyDeviceToHost));
    // the route: what the sample says (dense_pays), or argv[3] = lines | dense
    const bool dense = argc > 3 ? !strcmp(argv[3], "dense") : dense_pays(h_hist.data(), h_hist2.data());
    Codes h_codes = build_codes(dense ? h_hist2.data() : h_hist.data());
    h_codes.dense = dense ? 1u : 0u;
    CHECK(hipMemcpy(codes, &h_codes, sizeof h_codes, hipMemcpyHostToDevice));
    auto launch = [&](uint32_t *trace) {
        if (dense) hipLaunchKernelGGL((k_gzip_pieces<false, kDenseStep>), dim3(n_pieces), dim3(kThreads), 0, 0, text, n, 1u, codes, slots, sizes, trace);
        else hipLaunchKernelGGL((k_gzip_pieces<false, kProbeStep>), dim3(n_pieces), dim3(kThreads), 0, 0, text, n, 1u, codes, slots, sizes, trace);
    };
    hipEvent_t e0, e1, e2;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventCreate(&e2));
#if defined(RSQ_GZ_TRACE)
    unsigned long long *trace = nullptr;
    CHECK(hipMalloc(&trace, 64));
    CHECK(hipMemset(trace, 0, 64));
    launch(reinterpret_cast<uint32_t *>(trace));
    unsigned long long h_trace[8];
    CHECK(hipMemcpy(h_trace, trace, 64, hipMemcpyDeviceToHost));
    {
        static const char *names[8] = {"ring", "lines+list", "A1 hash", "A2 probes", "B count", "B emit", "flush", "eob+crc"};
        unsigned long long sum = 0;
        for (int i = 0; i < 8; ++i) sum += h_trace[i];
        fprintf(stderr, "clocks of thread 0 per phase (share of its workgroups' time):");
        for (int i = 0; i < 8; ++i) fprintf(stderr, "  %s %.1f%%", names[i], 100.0 * (double)h_trace[i] / (double)sum);
        fprintf(stderr, "\n");
    }
#endif
    float best = 1e30f, best_stored = 0;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        launch(nullptr);
        CHECK(hipEventRecord(e1, 0));
        hipLaunchKernelGGL(k_gzip_stored, dim3(n_pieces), dim3(kThreads), 0, 0, text, n, slots, sizes);
        CHECK(hipEventRecord(e2, 0));
        CHECK(hipEventSynchronize(e2));
        float ms = 0, ms2 = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipEventElapsedTime(&ms2, e1, e2));
        if (ms < best) best = ms, best_stored = ms2;
    }
    std::vector<uint32_t> h_sizes(n_pieces);
    CHECK(hipMemcpy(h_sizes.data(), sizes, n_pieces * 4, hipMemcpyDeviceToHost));
    uint64_t total = 0, stored = 0;
    for (uint64_t i = 0; i < n_pieces; ++i) {
        total += h_sizes[i];
        stored += h_sizes[i] == kHeaderBytes + 5u + std::min<uint64_t>(kPiece, n - i * kPiece) + kTrailerBytes;
    }
    // the first members (and the last one): inflate with zlib, compare with the text and with the host walk
    std::vector<uint8_t> slot(kSlot), host_slot(kSlot), back(kPiece + 16);
    int bad = 0;
    const uint64_t check[] = {0, 1, 2, n_pieces / 2, n_pieces - 1};
    std::vector<uint8_t> whole(n < ((size_t)64 << 20) ? n : 0);      // the host needs the text of the checked pieces: a piece of copy c is a stretch of `one` (wrapping)
    for (uint64_t piece : check) {
        if (piece >= n_pieces) continue;
        const uint32_t len = (uint32_t)std::min<uint64_t>(kPiece, n - piece * kPiece);
        std::vector<uint8_t> want(len);
        for (uint32_t i = 0; i < len; ++i) want[i] = one[(piece * kPiece + i) % one.size()];
        CHECK(hipMemcpy(slot.data(), slots + piece * kSlot, kSlot, hipMemcpyDeviceToHost));
        const uint32_t size = h_sizes[piece];
        z_stream z;
        memset(&z, 0, sizeof z);
        inflateInit2(&z, 31);
        z.next_in = slot.data() + kSlotPad;
        z.avail_in = size;
        z.next_out = back.data();
        z.avail_out = (uInt)back.size();
        const int rc = inflate(&z, Z_FINISH);
        const bool ok = rc == Z_STREAM_END && z.total_out == len && !memcmp(back.data(), want.data(), len);
        inflateEnd(&z);
        uint32_t host_size = piece_on_the_host(want.data(), len, &h_codes, host_slot.data(), nullptr, dense);
        if (!host_size) host_size = stored_piece_on_the_host(want.data(), len, host_slot.data());
        const bool same = host_size == size && !memcmp(host_slot.data() + kSlotPad, slot.data() + kSlotPad, size);
        if (!ok || !same) {
            ++bad;
            fprintf(stderr, "piece %llu: inflate %s (rc %d, %lu bytes of %u), host walk %s (%u vs %u bytes)\n", (unsigned long long)piece, ok ? "ok" : "WRONG", rc, z.total_out, len,
                    same ? "equal" : "DIFFERENT", host_size, size);
        }
    }
    uLongf z1 = compressBound(one.size()), z6 = z1;
    std::vector<uint8_t> zbuf(z1);
    compress2(zbuf.data(), &z1, one.data(), one.size(), 1);
    compress2(zbuf.data(), &z6, one.data(), one.size(), 6);
    printf("{\"text_bytes\": %llu, \"pieces\": %llu, \"kernel_ms\": %.3f, \"stored_kernel_ms\": %.3f, \"gbytes_per_s\": %.1f, \"ms_per_7p5_GB\": %.1f, \"members_bytes\": %llu, \"ratio\": %.3f, "
           "\"zlib1_ratio\": %.3f, \"zlib6_ratio\": %.3f, \"size_over_zlib1\": %.3f, \"stored_pieces\": %llu, \"checked_pieces_wrong\": %d, \"route\": \"%s\"}\n",
           (unsigned long long)n, (unsigned long long)n_pieces, best, best_stored, n / (best * 1e-3) / 1e9, 7.5e9 / (n / (best * 1e-3)) * 1e3, (unsigned long long)total, (double)n / total,
           (double)one.size() / z1, (double)one.size() / z6, ((double)total / n) / ((double)z1 / one.size()), (unsigned long long)stored, bad, dense ? "dense" : "lines");
    return bad ? 1 : 0;
}
