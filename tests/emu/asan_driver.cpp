// asan_driver.cpp — TEST-ONLY: the product's host front-end (csrc/host/frontend.cpp), its planner for the device entropy
// decoders, the compact transport writer and the oracle's front-end, run under AddressSanitizer + UBSan over JPEG files
// (the reference fuzzes its decoder, fuzz/; this is the sanitizer leg for the C / C++ restatements).
//   asan_driver file...      exit code 0 = no sanitizer report (decoding errors are expected outcomes, not failures)
// Built by tests/test_sanitizers.py:  g++ -fsanitize=address,undefined ... asan_driver.cpp frontend.cpp image_job.cpp + oracle/*.c
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../jpeg-decoder_amd/csrc/compact.hpp"
#include "../../jpeg-decoder_amd/csrc/host/frontend.hpp"
extern "C" {
#include "../../oracle/jpeg_oracle.h"
}

using namespace jpgpu::host;

struct CountingSink : RowSink {
    jpgpu_component comp[4];
    uint16_t qt[4][64];
    size_t rows[4] = {0, 0, 0, 0};
    uint64_t checksum = 0;
    std::vector<uint8_t> compact;
    void start(uint32_t index, const jpgpu_component &c, const uint16_t q[64]) override {
        comp[index & 3] = c;
        memcpy(qt[index & 3], q, 128);
        rows[index & 3] = 0;
    }
    void append_row(uint32_t index, const int16_t *coefficients, size_t len) override {
        rows[index & 3]++;
        for (size_t i = 0; i < len; i += 61) checksum = checksum * 31 + (uint16_t)coefficients[i];
        // every row through the compact transport writer as well (what the pipeline sends over PCIe)
        const size_t nblk = len / 64;
        compact.resize(jpgpu::compact_max_bytes(nblk));
        jpgpu::CompactWriter w(compact.data(), nblk, qt[index & 3]);
        w.add_blocks(coefficients, nblk);
        int cls = 0;
        checksum += w.finish(&cls) + (size_t)cls;
    }
    void finish(uint32_t, uint32_t) override {}
};

int main(int argc, char **argv) {
    unsigned ok = 0, failed = 0, planned = 0;
    for (int a = 1; a < argc; a++) {
        FILE *f = fopen(argv[a], "rb");
        if (!f) continue;
        std::vector<uint8_t> data;
        uint8_t buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof(buf), f)) > 0) data.insert(data.end(), buf, buf + n);
        fclose(f);
        try {  // Decoder::decode through the product's front-end
            Frontend fe(data.data(), data.size());
            fe.set_max_decoding_buffer_size(64u << 20);
            CountingSink sink;
            fe.decode_to(sink);
            ok++;
        } catch (const DecodeError &) {
            failed++;
        }
        try {  // the planner of the device entropy route (walks every marker without decoding)
            Frontend fe(data.data(), data.size(), Frontend::Borrowed{});
            std::vector<PlannedScan> plans;
            if (fe.plan_device_scans(plans)) planned++;
        } catch (const DecodeError &) {
        }
        orc_result res;  // the oracle's own front-end and pixel path on the same bytes
        memset(&res, 0, sizeof(res));
        orc_decode(data.data(), data.size(), 0, 0, -1, 0, &res);
        orc_free_result(&res);
    }
    printf("%u decoded, %u refused, %u eligible for the device entropy route\n", ok, failed, planned);
    return 0;
}
