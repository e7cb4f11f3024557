// Fuzz harness for strainscan_amd/csrc/ss_pgz.hip, built with g++ -fsanitize=address,undefined by
// tests/test_abi_and_host.py::test_threaded_gunzip_fuzz_sanitized.  usage: pgz_fuzz file.gz expected.txt iterations
// Every iteration damages the gzip image (byte flips, a bit flip, truncation, a zeroed stretch) and inflates it with
// 2..5 threads: the result must be refused, or be the expected text; the sanitizers watch every access.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "../strainscan_amd/csrc/ss_pgz.hip"

static std::vector<uint8_t> slurp(const char *p)
{
    std::vector<uint8_t> v;
    FILE *f = fopen(p, "rb");
    if (!f) { perror(p); exit(2); }
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) v.insert(v.end(), buf, buf + n);
    fclose(f);
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    const std::vector<uint8_t> gz = slurp(argv[1]), want = slurp(argv[2]);
    const int iters = atoi(argv[3]);
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    auto rnd = [&] { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    int accepted = 0, refused = 0;
    for (int it = -1; it < iters; it++) {
        std::vector<uint8_t> img = gz;
        if (it >= 0) {
            switch (rnd() % 5) {
            case 0: for (int k = 0; k < 1 + (int)(rnd() % 4); k++) img[rnd() % img.size()] ^= (uint8_t)(1 + rnd() % 255); break;
            case 1: img[rnd() % img.size()] ^= (uint8_t)(1u << (rnd() % 8)); break;
            case 2: img.resize(18 + rnd() % (img.size() - 18)); break;
            case 3: { const size_t a = rnd() % img.size(), n = 1 + rnd() % 5000; for (size_t i = a; i < img.size() && i < a + n; i++) img[i] = 0; break; }
            default: { const size_t a = rnd() % img.size(), n = 1 + rnd() % 300; for (size_t i = a; i < img.size() && i < a + n; i++) img[i] = (uint8_t)rnd(); break; }
            }
        }
        char *text = nullptr;
        uint64_t len = 0;
        const unsigned threads = 2 + (unsigned)(rnd() % 4);
        if (ss::parallel_gunzip(img.data(), img.size(), threads, 1ull << 34, &text, &len)) {
            if (len != want.size() || memcmp(text, want.data(), len) != 0) { fprintf(stderr, "iteration %d: WRONG TEXT accepted\n", it); return 1; }
            free(text);
            accepted++;
        } else {
            if (it < 0) { fprintf(stderr, "the undamaged file was refused\n"); return 1; }
            refused++;
        }
    }
    printf("pgz_fuzz ok: %d accepted (all identical to the expected text), %d refused\n", accepted, refused);
    return 0;
}
