// host_bench.cpp — single-thread cost of the host half of the pipeline on one JPEG file:
//   g++ -O3 -std=c++17 -I. tools/host_bench.cpp jpeg-decoder_amd/csrc/host/frontend.cpp jpeg-decoder_amd/csrc/image_job.cpp -o /tmp/host_bench
//   /tmp/host_bench file.jpg [repeats]
// Prints ms per image for: entropy decoding alone (rows dropped), + dense staging (memcpy + range scan), + compact staging; and, for a
// progressive file, where the entropy decoding time goes scan by scan (the clock read whenever the front-end reports what a scan changed).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>

#include "jpeg-decoder_amd/csrc/compact.hpp"
#include "jpeg-decoder_amd/csrc/host/frontend.hpp"

using namespace jpgpu::host;

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct NullSink : RowSink {
    void start(uint32_t, const jpgpu_component &, const uint16_t *) override {}
    void append_row(uint32_t, const int16_t *, size_t) override {}
    void finish(uint32_t, uint32_t) override {}
};
struct DenseSink : RowSink {
    std::vector<int16_t> buf[4];
    size_t w[4] = {0, 0, 0, 0};
    uint16_t q[4][64];
    int rc = 0;
    void start(uint32_t i, const jpgpu_component &c, const uint16_t *qt) override {
        buf[i].resize((size_t)c.block_width * c.block_height * 64);
        w[i] = 0;
        memcpy(q[i], qt, 128);
    }
    void append_row(uint32_t i, const int16_t *co, size_t len) override {
        memcpy(buf[i].data() + w[i], co, len * 2);
        w[i] += len;
    }
    void finish(uint32_t i, uint32_t) override { rc += jpgpu_range_class(buf[i].data(), buf[i].size(), q[i]); }
};
struct CompactSink : RowSink {
    std::vector<uint8_t> buf[4];
    std::unique_ptr<jpgpu::CompactWriter> wr[4];
    size_t bytes = 0;
    uint16_t q[4][64];
    void start(uint32_t i, const jpgpu_component &c, const uint16_t *qt) override {
        const size_t nb = (size_t)c.block_width * c.block_height;
        buf[i].resize(jpgpu::compact_max_bytes(nb));
        memcpy(q[i], qt, 128);
        wr[i].reset(new jpgpu::CompactWriter(buf[i].data(), nb, q[i]));
    }
    void append_row(uint32_t i, const int16_t *co, size_t len) override { wr[i]->add_blocks(co, len / 64); }
    void finish(uint32_t i, uint32_t) override {
        int rc;
        bytes += wr[i]->finish(&rc);
    }
};

// progressive files: a sink that reads the clock whenever the front-end reports a finished scan (RowSink::scan_finished, once per
// component of the scan)
struct ScanClockSink : NullSink {
    std::vector<double> at;
    std::vector<uint32_t> comp;
    void scan_finished(uint32_t slot) override {
        at.push_back(now_ms());
        comp.push_back(slot);
    }
};

int main(int argc, char **argv) {
    if (argc < 2) return 1;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 1;
    std::vector<uint8_t> data;
    uint8_t tmp[65536];
    size_t n;
    while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) data.insert(data.end(), tmp, tmp + n);
    fclose(f);
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    DenseSink ds;
    CompactSink cs;
    for (int mode = 0; mode < 3; mode++) {
        double best = 1e9;
        for (int r = 0; r < reps; r++) {
            const double t0 = now_ms();
            Frontend fe(data.data(), data.size());
            NullSink ns;
            cs.bytes = 0;
            if (mode == 0) fe.decode_to(ns);
            else if (mode == 1) fe.decode_to(ds);
            else fe.decode_to(cs);
            best = std::min(best, now_ms() - t0);
        }
        printf("%s: %.3f ms\n", mode == 0 ? "entropy only" : mode == 1 ? "entropy + dense staging + range scan" : "entropy + compact staging", best);
    }
    printf("compact bytes %zu\n", cs.bytes);
    {  // scan by scan (median of `reps` runs per interval); sequential files report nothing here
        std::vector<std::vector<double>> dt;
        std::vector<uint32_t> comp;
        double total = 0;
        for (int r = 0; r < reps; r++) {
            ScanClockSink sc;
            const double t0 = now_ms();
            Frontend fe(data.data(), data.size());
            fe.decode_to(sc);
            const double t1 = now_ms();
            if (sc.at.empty()) break;
            if (dt.empty()) dt.resize(sc.at.size() + 1), comp = sc.comp;
            if (sc.at.size() + 1 != dt.size()) break;
            for (size_t k = 0; k < sc.at.size(); k++) dt[k].push_back(sc.at[k] - (k ? sc.at[k - 1] : t0));
            dt.back().push_back(t1 - sc.at.back());
            total += t1 - t0;
        }
        if (!dt.empty() && !dt[0].empty()) {
            printf("progressive: %.3f ms per image (mean); per reported scan x component (median ms):\n", total / (double)dt[0].size());
            double sum = 0;
            for (size_t k = 0; k < dt.size(); k++) {
                std::sort(dt[k].begin(), dt[k].end());
                const double m = dt[k][dt[k].size() / 2];
                sum += m;
                if (k + 1 < dt.size()) printf("  report %2zu  component %u  %8.4f ms\n", k, comp[k], m);
                else printf("  after the last report (finishing rows)  %8.4f ms\n", m);
            }
            printf("  sum of medians %.3f ms\n", sum);
        }
    }
    return 0;
}
