// ASan + UBSan over the two files of the device JPEG decoder that read untrusted bytes on the host (jpeg_parse.cpp,
// jpeg_prog.cpp), without a GPU and without the rest of the library: every file given on the command line is parsed and --
// progressive ones -- entropy-decoded as it is, in every truncation, and under `iters` random mutations each.
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include \
//       -Ifennec_amd/csrc tools/fuzz_jpeg_host.cpp fennec_amd/csrc/jpeg_parse.cpp fennec_amd/csrc/jpeg_prog.cpp -o /tmp/fuzz_jpeg_host
//   /tmp/fuzz_jpeg_host 20000 file1.jpg file2.jpg ...        (tools/fuzz_jpeg_host.sh makes the files with Pillow and runs it)
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "common.hpp"

namespace fnx {
void set_error(const char *, ...) {}
}  // namespace fnx

static long n_ok = 0, n_unsupported = 0, n_invalid = 0;

static void one(const std::vector<uint8_t> &d)
{
    fnx::JpegFile f;
    int rc = fnx::jpeg_parse(d.data(), d.size(), &f);
    if (rc == 0 && f.progressive) {
        const unsigned long long nblk = static_cast<unsigned long long>(f.mx) * f.my * f.nslots;
        if (nblk > (1ull << 22)) { n_unsupported++; return; }               // (the library sizes by the file's length first; keep the harness's memory bounded)
        std::vector<int16_t> coef(64 * static_cast<size_t>(nblk), 0);
        rc = fnx::jpeg_progressive_coefficients(d.data(), d.size(), &f, coef.data());
    } else if (rc == 0) {
        std::vector<uint8_t> out(d.size() - f.scan + 64);
        std::vector<uint32_t> rst;
        size_t nb = 0;
        rc = fnx::jpeg_unstuff(d.data(), d.size(), f, out.data(), &nb, &rst);
    }
    if (rc == 0) n_ok++; else if (rc == FNX_ERR_UNSUPPORTED) n_unsupported++; else n_invalid++;
}

int main(int argc, char **argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s iters file...\n", argv[0]); return 2; }
    const long iters = std::atol(argv[1]);
    std::mt19937_64 rng(12345);
    for (int a = 2; a < argc; a++) {
        FILE *fp = std::fopen(argv[a], "rb");
        if (!fp) { std::perror(argv[a]); return 2; }
        std::vector<uint8_t> g;
        uint8_t buf[65536];
        size_t k;
        while ((k = std::fread(buf, 1, sizeof buf, fp)) > 0) g.insert(g.end(), buf, buf + k);
        std::fclose(fp);
        one(g);
        for (size_t cut = 0; cut < g.size(); cut += (cut < 700 ? 1 : 13)) one(std::vector<uint8_t>(g.begin(), g.begin() + cut));
        for (long it = 0; it < iters; it++) {
            std::vector<uint8_t> c = g;
            const int nm = 1 + static_cast<int>(rng() % 4);
            for (int m = 0; m < nm; m++) {
                const size_t at = 2 + rng() % (c.size() - 2);
                switch (rng() % 4) {
                case 0: c[at] = static_cast<uint8_t>(rng()); break;
                case 1: c[at] = 0xff; break;
                case 2: c[at] ^= static_cast<uint8_t>(1u << (rng() % 8)); break;
                default: if (at + 1 < c.size()) { c[at] = 0xff; c[at + 1] = static_cast<uint8_t>(0xc0 + rng() % 0x30); } break;
                }
            }
            one(c);
        }
    }
    std::printf("fuzz_jpeg_host: %ld decoded, %ld unsupported, %ld invalid; no sanitizer report\n", n_ok, n_unsupported, n_invalid);
    return 0;
}
