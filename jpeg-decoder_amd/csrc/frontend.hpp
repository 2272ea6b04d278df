// frontend.hpp -- host JPEG front end: header model, Huffman tables, entropy decoder.
//
// Pure host C++ (no HIP): fills the full-frame blocl16_t planes the GPU back end consumes.
// It reproduces the reference's accept/reject decisions and coefficient values (src/entete.c,
// src/vld.c, src/bitstream.c, src/baseline.c, src/progressive.c) with its own machinery: the file is
// read from memory, scan data is un-stuffed once per scan, Huffman symbols are resolved by a
// 9-bit lookup table with a tree walk only for longer codes, and coefficients land directly in
// contiguous planes.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace jpegfe {

struct HuffTable {
   // Mirror of the breadth-first tree of remplir_huffman (src/entete.c:452-507): canonical codes, and
   // nodes that were created but never given a symbol stay leaves with symbol 0 (calloc), which
   // is what a corrupt stream decodes to in the reference.
   // A table whose shortest code is 15 or 16 bits long makes the breadth-first build expand every node above
   // that depth (up to 2^17 - 1 nodes), so node indices are 32-bit.
   struct Node { int32_t child[2]; uint8_t sym; };
   std::vector<Node> nodes;
   uint16_t fast[512];   // (length << 8) | symbol for codes of <= 9 bits, else 0
   int32_t deep[512];    // node reached after 9 bits when fast[] is 0
   // AC coefficient in one look-up: index = the next 10 bits; (value << 16) | (zero run << 8) | (code length + value bits)
   // when code and value bits fit into those 10 bits and the symbol carries a value (SSSS 1..10), else 0
   int32_t acfast[1024];
   bool present = false;
   int build(const uint8_t counts[16], const uint8_t *symbols, unsigned nsym);   // 0 or ERR_HUFF_BAD
};

struct Component { uint8_t idc = 0, hs = 0, vs = 0, idhdc = 0, idhac = 0, idq = 0; };

struct FastBits;
struct BitReader {
   const uint8_t *buf = nullptr;   // un-stuffed scan bytes + the marker's 0xFF + zero guard
   uint64_t pos = 0;               // bits consumed
   uint64_t limit = 0;             // consuming bit <limit-1> (the 8th bit of the marker's 0xFF) is an error
   bool fast_allowed = true;       // blocks far from the end of the scan may use the unchecked reader (frontend.cpp FastBits)
   bool bmi2 = false;              // ... and the PDEP form of the refinement scans (x86-64 with BMI2, found at run time)
};

struct Decoder {
   // input
   const uint8_t *data = nullptr;
   size_t len = 0, pos = 0;
   std::vector<uint8_t> owned;
   // img_t (include/img.h:88-103)
   uint16_t width = 0, height = 0;
   uint8_t precision = 0, nbcomp = 0;
   Component comp[3];
   uint8_t ordre[3] = { 0, 0, 0 };
   bool app0_done = false, sof_done = false, dqt_done = false, dht_done = false, sos_done = false, eoi_done = false;
   uint8_t num_sof = 0;
   char jfif[5] = { 0, 0, 0, 0, 0 };
   uint8_t ver_x = 0, ver_y = 0, ss = 0, se = 0, ah = 0, al = 0;
   bool qt_present[4] = { false, false, false, false };
   uint8_t qt_precision[4] = { 0, 0, 0, 0 };
   uint16_t qt[4][64];
   HuffTable hdc[4], hac[4];
   std::vector<std::string> comments;
   uint8_t max_h = 0, max_v = 0;
   uint32_t nbmcuH = 0, nbmcuV = 0;
   uint8_t nb_scan_comp = 0;
   bool too_many_comps = false;
   // decoder state
   int16_t *plane[3] = { nullptr, nullptr, nullptr };   // zeroed full-frame planes, owned by the caller (pinned when a GPU is present)
   bool dirty[3] = { false, false, false };
   // What the scans since the last upload changed in a component's plane: only coefficient 0 of its blocks (DC scans,
   // src/progressive.c:98-126) or anything.  With dc_pack[c] attached, DC scans also leave every block's coefficient 0 in
   // dc_pack[c][block index] -- the "dirty spectral band" a per-scan hand-off uploads instead of the whole plane.
   enum : uint8_t { kClean = 0, kDcOnly = 1, kWhole = 2 };
   uint8_t dirty_kind[3] = { kClean, kClean, kClean };
   int16_t *dc_pack[3] = { nullptr, nullptr, nullptr };
   // dc_in_pack: DC scans keep coefficient 0 of a component with a dc_pack ONLY there (2 bytes per block, dense) instead of
   // touching one 128-byte line per block of the plane for it; plane[c][64 * b] then stays 0 until merge_dc_into_planes().
   // The GPU side puts the packed values in place on the device.  dc_dirty[c]: a DC scan changed dc_pack[c] since the caller
   // last cleared the flag.
   bool dc_in_pack = false;
   bool dc_dirty[3] = { false, false, false };
   void merge_dc_into_planes(int only = -1);   // only: one component, -1 = all
   unsigned scans_decoded = 0;
   bool dequant_on_store = false;     // baseline only: store coef * qtable (int16 wrap) like decode_bloc_baseline's iquant (src/baseline.c:66)
   uint32_t mcu_rows_done = 0;        // complete MCU rows of the last baseline scan (partial output on error)
   // baseline scans: called with the number of complete MCU rows every <row_hook_every> rows (not after the last one), so
   // that the caller can hand finished rows on while the rest is being decoded; a non-zero return ends the scan with it
   int (*row_hook)(void *ctx, uint32_t mcu_rows_done) = nullptr;
   void *row_hook_ctx = nullptr;
   uint32_t row_hook_every = 0;
   std::vector<uint8_t> scanbuf;
   std::string err;

   int open();                 // decode_entete(first pass)
   int next_header();          // decode_entete(later pass)
   int decode_scan();          // entropy-decode the pending scan + finir_octet
   void dequantise_planes();   // iquant(0, 63) over every block with the current tables (dense; progressive batch path)
   int get_composante(unsigned k) const;
   size_t plane_blocks(int c) const { return (size_t)nbmcuH * comp[c].hs * (size_t)nbmcuV * comp[c].vs; }

 private:
   int fail(int code, const char *fmt, ...);
   int getc();
   int get16();
   int marker();
   int seg_app0();
   int seg_com();
   int seg_sof();
   int seg_dqt();
   int seg_dht();
   int seg_sos();
   int check_app0();
   int check_baseline();
   int check_progressive();
   int header_loop(bool first);
   void geometry();
   int begin_scan(BitReader &br, size_t &scan_begin, size_t &scan_end, size_t &nbytes);
   int finish_scan(const BitReader &br, size_t scan_begin, size_t scan_end, size_t nbytes);
   int scan_baseline(BitReader &br);
   int scan_progressive_dc(BitReader &br);
   int scan_progressive_ac(BitReader &br);
   // R: CheckedBits / FastBits (frontend.cpp)
   template <class R> int ac_first(R &r, const HuffTable &t, int16_t *blk, uint16_t &skip, const uint16_t *q);
   template <class R> int ac_refine(R &r, const HuffTable &t, int16_t *blk, uint16_t &skip);
   template <class R> int ac_refine_walk(R &r, const HuffTable &t, int16_t *blk, uint16_t &skip, unsigned idx);
   int ac_refine_bmi2(FastBits &r, const HuffTable &t, int16_t *blk, uint16_t &skip);
   template <class R> int refine_to_band_end(R &r, int16_t *blk, unsigned idx);
};

}  // namespace jpegfe
