// tools/fuzz_frontend.cpp -- host-only robustness driver for the front end: random damage (bit flips, random bytes,
// truncation, insertions, deletions) on the files of a directory, every scan decoded.  Build with
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -Ijpeg-decoder_amd/csrc -Iinclude tools/fuzz_frontend.cpp jpeg-decoder_amd/csrc/frontend.cpp
// usage: fuzz <dir> <iterations>.  300 000 iterations over tests/golden/jpeg: no sanitizer finding, no hang.
// Every variant is decoded three times: as shipped (blocks far from the end of a scan go through the unchecked register
// bit reader and the bit-map refinement, with PDEP where the CPU has BMI2), the same without BMI2, and with every block
// through the checked reader that mirrors the reference's read_bit; status, message and every coefficient must be equal.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include <dirent.h>
#include "frontend.hpp"
static uint64_t s = 88172645463325252ull;
static uint64_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
extern "C" void jpeggpu_internal_vld_force_checked(int mode);   // 0 default, 1 checked reader everywhere, 2 default without BMI2
struct Result { int rc = 0; std::string err; std::vector<std::vector<int16_t>> planes; long scans = 0; uint32_t rows = 0; };
static Result decode(const std::vector<uint8_t> &b, bool dequant, int mode) {
   jpeggpu_internal_vld_force_checked(mode);
   Result r; r.planes.resize(3);
   jpegfe::Decoder dec; dec.data = b.data(); dec.len = b.size();
   int rc = dec.open();
   if (rc == 0) {
      for (int c = 0; c < dec.nbcomp; c++) { size_t nb = dec.plane_blocks(c); if (nb > (1u << 22)) { rc = -100; break; } r.planes[c].assign(nb * 64 + 64, 0); dec.plane[c] = r.planes[c].data(); }
      dec.dequant_on_store = dequant && dec.num_sof == 0;
      int guard = 0;
      while (rc == 0 && !dec.eoi_done && guard++ < 2000) { rc = dec.decode_scan(); r.scans++; if (rc == 0 && !dec.eoi_done) rc = dec.next_header(); }
      if (rc == 0 && dec.num_sof) dec.dequantise_planes();
   }
   r.rc = rc; r.err = dec.err; r.rows = dec.mcu_rows_done;
   return r;
}
int main(int argc, char **argv) {
   std::vector<std::vector<uint8_t>> files, large;   // large: scans long enough for the unchecked reader to be used
   DIR *d = opendir(argv[1]); struct dirent *e;
   while ((e = readdir(d))) { std::string n = e->d_name; if (n.size() < 4 || (n.find(".jp") == std::string::npos)) continue;
      std::string p = std::string(argv[1]) + "/" + n; FILE *f = fopen(p.c_str(), "rb"); std::vector<uint8_t> b; int c; while ((c = fgetc(f)) != EOF) b.push_back((uint8_t)c); fclose(f);
      if (b.size() < 100000) { files.push_back(b); if (b.size() > 2000) large.push_back(b); } }
   closedir(d);
   long iters = atol(argv[2]); long ok = 0, scans = 0; int hist[64] = {0};
   for (long it = 0; it < iters; it++) {
      const bool big = (it & 1) && !large.empty();
      std::vector<uint8_t> b = big ? large[rnd() % large.size()] : files[rnd() % files.size()];
      int nm = 1 + rnd() % 4;
      for (int k = 0; k < nm; k++) {
         // a large file is damaged behind its headers most of the time, so that decoding gets as far as the scans
         const size_t lo = big && (rnd() % 8) ? b.size() / 8 : 0;
         switch (rnd() % 5) {
         case 0: b[lo + rnd() % (b.size() - lo)] ^= (uint8_t)(1u << (rnd() % 8)); break;
         case 1: b[lo + rnd() % (b.size() - lo)] = (uint8_t)rnd(); break;
         case 2: if (b.size() > 8) { size_t cut = 4 + rnd() % (b.size() - 6); b.resize(cut); b.push_back(0xFF); b.push_back(0xD9); } break;
         case 3: { size_t p = lo + rnd() % (b.size() - lo); b.insert(b.begin() + p, (uint8_t)rnd()); } break;
         case 4: if (b.size() > 10) { size_t p = rnd() % (b.size() - 4); b.erase(b.begin() + p); } break;
         }
      }
      const Result fast = decode(b, (it & 2) != 0, 0), ref = decode(b, (it & 2) != 0, 1), plain = decode(b, (it & 2) != 0, 2);
      for (const Result *o : { &fast, &plain })
         if (o->rc != ref.rc || o->err != ref.err || o->planes != ref.planes || o->scans != ref.scans || o->rows != ref.rows) {
            printf("MISMATCH (%s) at iteration %ld: rc %d / %d, '%s' / '%s', scans %ld / %ld\n", o == &fast ? "default" : "no BMI2", it, o->rc, ref.rc,
                   o->err.c_str(), ref.err.c_str(), o->scans, ref.scans);
            FILE *f = fopen("fuzz_mismatch.jpg", "wb"); if (f) { fwrite(b.data(), 1, b.size(), f); fclose(f); }
            return 1;
         }
      const int rc = fast.rc; scans += fast.scans;
      if (rc == 0) ok++;
      hist[rc < 0 ? 63 : (rc > 62 ? 62 : rc)]++;
   }
   printf("iterations %ld, fully decoded %ld, scans %ld\nstatus histogram:", iters, ok, scans);
   for (int i = 0; i < 64; i++) if (hist[i]) printf(" %d:%d", i == 63 ? -1 : i, hist[i]);
   printf("\n");
   return 0;
}
