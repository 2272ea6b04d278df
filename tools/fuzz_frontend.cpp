// tools/fuzz_frontend.cpp -- host-only robustness driver for the front end: random damage (bit flips, random bytes,
// truncation, insertions, deletions) on the files of a directory, every scan decoded.  Build with
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -Ijpeg-decoder_amd/csrc -Iinclude tools/fuzz_frontend.cpp jpeg-decoder_amd/csrc/frontend.cpp
// usage: fuzz <dir> <iterations>.  300 000 iterations over tests/golden/jpeg: no sanitizer finding, no hang.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include <dirent.h>
#include "frontend.hpp"
static uint64_t s = 88172645463325252ull;
static uint64_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
int main(int argc, char **argv) {
   std::vector<std::vector<uint8_t>> files;
   DIR *d = opendir(argv[1]); struct dirent *e;
   while ((e = readdir(d))) { std::string n = e->d_name; if (n.size() < 4 || (n.find(".jp") == std::string::npos)) continue;
      std::string p = std::string(argv[1]) + "/" + n; FILE *f = fopen(p.c_str(), "rb"); std::vector<uint8_t> b; int c; while ((c = fgetc(f)) != EOF) b.push_back((uint8_t)c); fclose(f);
      if (b.size() < 100000) files.push_back(b); }
   closedir(d);
   long iters = atol(argv[2]); long ok = 0, scans = 0; int hist[64] = {0};
   for (long it = 0; it < iters; it++) {
      std::vector<uint8_t> b = files[rnd() % files.size()];
      int nm = 1 + rnd() % 4;
      for (int k = 0; k < nm; k++) {
         switch (rnd() % 5) {
         case 0: b[rnd() % b.size()] ^= (uint8_t)(1u << (rnd() % 8)); break;
         case 1: b[rnd() % b.size()] = (uint8_t)rnd(); break;
         case 2: if (b.size() > 8) { size_t cut = 4 + rnd() % (b.size() - 6); b.resize(cut); b.push_back(0xFF); b.push_back(0xD9); } break;
         case 3: { size_t p = rnd() % b.size(); b.insert(b.begin() + p, (uint8_t)rnd()); } break;
         case 4: if (b.size() > 10) { size_t p = rnd() % (b.size() - 4); b.erase(b.begin() + p); } break;
         }
      }
      jpegfe::Decoder dec; dec.data = b.data(); dec.len = b.size();
      int rc = dec.open();
      std::vector<std::vector<int16_t>> planes(3);
      if (rc == 0) {
         for (int c = 0; c < dec.nbcomp; c++) { size_t nb = dec.plane_blocks(c); if (nb > (1u << 22)) { rc = -100; break; } planes[c].assign(nb * 64 + 64, 0); dec.plane[c] = planes[c].data(); }
         dec.dequant_on_store = (it & 1) && dec.num_sof == 0;
         int guard = 0;
         while (rc == 0 && !dec.eoi_done && guard++ < 2000) { rc = dec.decode_scan(); scans++; if (rc == 0 && !dec.eoi_done) rc = dec.next_header(); }
         if (rc == 0 && dec.num_sof) dec.dequantise_planes();
      }
      if (rc == 0) ok++;
      hist[rc < 0 ? 63 : (rc > 62 ? 62 : rc)]++;
   }
   printf("iterations %ld, fully decoded %ld, scans %ld\nstatus histogram:", iters, ok, scans);
   for (int i = 0; i < 64; i++) if (hist[i]) printf(" %d:%d", i == 63 ? -1 : i, hist[i]);
   printf("\n");
   return 0;
}
