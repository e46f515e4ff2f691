// Mutation fuzz of the segment-image parsers (vecgo_amd/csrc/vg_segment_layout.hpp: every read of untrusted bytes that
// vg_segment_open_flat / _diskann do before a section goes to the device), compiled by plain g++ with
// -fsanitize=address,undefined (no HIP, no GPU).  For every mutated image: parse (with and without the checksum test); if the
// parser accepts it, check that every section it declares lies inside the image and READ every byte of it — the image sits in a
// heap block of exactly its size, so a section that sticks out by one byte is an AddressSanitizer report.
//   usage: segment_fuzz ITERATIONS SEED image...      (images: tests/segfile.py's writers, flat and DiskANN)
// VERDICT r05 weak 15 / next 8.  Reference formats: internal/segment/diskann/format.go:49-79, internal/segment/flat/format.go.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "vg_segment_layout.hpp"

using namespace vg::seglayout;

static uint64_t rng_state;
static uint64_t rnd()
{  // splitmix64
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static volatile uint64_t sink;
static int touch(const uint8_t *data, uint64_t len, const Section &s, const char *what)
{
    if (!s.present) return 0;
    if (s.off > len || s.bytes > len - s.off) {
        fprintf(stderr, "section %s [%llu, +%llu) outside an image of %llu bytes\n", what, (unsigned long long)s.off,
                (unsigned long long)s.bytes, (unsigned long long)len);
        abort();
    }
    uint64_t acc = 0;
    for (uint64_t i = 0; i < s.bytes; i++) acc += data[s.off + i];
    sink += acc;
    return 1;
}

int main(int argc, char **argv)
{
    if (argc < 4) {
        fprintf(stderr, "usage: segment_fuzz ITERATIONS SEED image...\n");
        return 2;
    }
    const long iterations = atol(argv[1]);
    rng_state = strtoull(argv[2], nullptr, 10);
    std::vector<std::vector<uint8_t>> images;
    for (int i = 3; i < argc; i++) {
        FILE *f = fopen(argv[i], "rb");
        if (!f) {
            perror(argv[i]);
            return 2;
        }
        std::vector<uint8_t> b;
        uint8_t buf[65536];
        size_t got;
        while ((got = fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + got);
        fclose(f);
        images.push_back(b);
    }
    // extreme values for the 64-bit offsets and 32-bit counts of a header
    const uint64_t extremes[] = {0, 1, 7, 8, 0xFFFFFFFFull, 0x100000000ull, 0x7FFFFFFFFFFFFFFFull, 0x8000000000000000ull,
                                 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFFFFFFFFF8ull, 0x2000000000000000ull, 0x5555555555555556ull};
    long accepted = 0, rejected = 0, sections = 0, pristine_ok = 0;
    for (long it = 0; it < iterations; it++) {
        const bool pristine = it < static_cast<long>(images.size());  // the first pass: every image unmodified
        const std::vector<uint8_t> &src = images[pristine ? static_cast<size_t>(it) : rnd() % images.size()];
        uint64_t len = src.size();
        const int muts = pristine ? 0 : 1 + static_cast<int>(rnd() % 4);
        if (muts && rnd() % 8 == 0) len = rnd() % (len + 1);           // truncation (header included)
        uint8_t *img = static_cast<uint8_t *>(malloc(len ? len : 1));  // exactly `len` bytes: one past the end is poisoned
        if (len) memcpy(img, src.data(), len);
        const uint64_t header = len < 160 ? len : 160;
        for (int m = 0; m < muts && len; m++) {
            switch (rnd() % 5) {
            case 0:  // any byte
                img[rnd() % len] = static_cast<uint8_t>(rnd());
                break;
            case 1:  // a header byte
                if (header) img[rnd() % header] = static_cast<uint8_t>(rnd());
                break;
            case 2: {  // a 64-bit header field <- an extreme, or the image size +- a little
                if (header < 8) break;
                const uint64_t at = (rnd() % (header / 8)) * 8;
                uint64_t v = extremes[rnd() % (sizeof extremes / sizeof extremes[0])];
                if (rnd() % 3 == 0) v = len + (rnd() % 33) - 16;
                memcpy(img + at, &v, 8);
                break;
            }
            case 3: {  // a 32-bit header field
                if (header < 4) break;
                const uint64_t at = (rnd() % (header / 4)) * 4;
                const uint32_t v = rnd() % 2 ? static_cast<uint32_t>(extremes[rnd() % 12]) : static_cast<uint32_t>(rnd());
                memcpy(img + at, &v, 4);
                break;
            }
            default:  // a bit
                img[rnd() % len] ^= static_cast<uint8_t>(1u << (rnd() % 8));
            }
        }
        for (int verify = 0; verify < 2; verify++) {
            FlatLayout fl;
            DiskLayout dl;
            Error e1, e2;
            if (parse_flat(img, len, verify != 0, fl, e1) == VG_OK) {
                accepted++;
                if (!muts) pristine_ok++;
                sections += touch(img, len, fl.sq_bounds, "sq_bounds") + touch(img, len, fl.pq_scales_offsets, "pq_scales_offsets") +
                            touch(img, len, fl.pq_codebooks, "pq_codebooks") + touch(img, len, fl.codes, "codes") +
                            touch(img, len, fl.vectors, "vectors") + touch(img, len, fl.centroids, "centroids") +
                            touch(img, len, fl.part_offsets, "part_offsets");
            } else if (parse_diskann(img, len, verify != 0, dl, e2) == VG_OK) {
                accepted++;
                if (!muts) pristine_ok++;
                sections += touch(img, len, dl.vectors, "vectors") + touch(img, len, dl.graph, "graph") +
                            touch(img, len, dl.pq_codes, "pq_codes") + touch(img, len, dl.pq_scales_offsets, "pq_scales_offsets") +
                            touch(img, len, dl.pq_codebooks, "pq_codebooks") + touch(img, len, dl.rabitq_codes, "rabitq_codes") +
                            touch(img, len, dl.int4_params, "int4_params") + touch(img, len, dl.int4_codes, "int4_codes");
            } else {
                rejected++;
                if (!muts) {
                    fprintf(stderr, "an unmodified image was rejected: %s / %s\n", e1.text.c_str(), e2.text.c_str());
                    abort();
                }
            }
        }
        free(img);
    }
    printf("segment_fuzz: %ld images, %ld parses accepted (%ld of unmodified images), %ld rejected, %ld sections read in full\n",
           iterations, accepted, pristine_ok, rejected, sections);
    return 0;
}
