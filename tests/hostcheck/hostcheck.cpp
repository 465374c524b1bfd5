// hostcheck.cpp -- the product's HOST-SIDE parsers under AddressSanitizer + UBSan, on the CPU (TEST INFRASTRUCTURE).
//
// GPU AddressSanitizer is not available on the pool; the code that parses UNTRUSTED container bytes, though, is plain C++ that
// needs no device: this program compiles the very text liborcgpu.so is built from -- orc_rust_amd/csrc/orcgpu_meta.inc
// (protobuf wire reader, PostScript / Footer / StripeFooter / RowIndex parsing, type-tree validation, row-index entry splitting,
// shard weights and deal), orcgpu_predicate.inc (row-group statistics, Bloom filters), orcgpu_selection.inc (RowSelection
// stepping), orcgpu_tz.inc (TZif / POSIX rule parsing) -- with `g++ -fsanitize=address,undefined` and walks files the way
// orcgpu_reader_open_* / reader_advance_stripe do.  Compressed sections are expanded with the CPU oracle's codecs (the product
// expands them on the GPU): the oracle is the checker's tool here, as everywhere under tests/.
//
//   hostcheck walk FILE                 one line: status, rows, stripes, types, streams, index entries
//   hostcheck fuzz FILE SEED N          the clean walk, then N seeded mutations (truncations; bit flips and byte overwrites in the
//                                       last 16 KiB and in every stripe footer); prints how many walks ended in which status
//   hostcheck tz FILE                   parses FILE as a TZif file (and its POSIX footer rule)
//   hostcheck select SEED N             N random RowSelections stepped over random stripes: the invariants of mod.rs:302-365
// Exit code 0 whenever the parsers RETURNED (with whatever status); a sanitizer report aborts with its own exit code.
#include <cinttypes>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <string>
#include <vector>

#include "../../include/orcgpu.h"
#include "../../orc_rust_amd/csrc/orcgpu_meta.inc"
struct SelBatch {  // (device/select_kernels.hip: what the selection kernel takes per selected batch)
  uint64_t start;
  uint32_t len;
  uint32_t pad;
};
#include "../../orc_rust_amd/csrc/orcgpu_selection.inc"
#include "../../orc_rust_amd/csrc/orcgpu_tz.inc"

extern "C" {
int oo_stream_decompress(const uint8_t* src, size_t n, int compression_kind, size_t block_size, uint8_t** out, size_t* out_len);
void oo_free(void* p);
}

using namespace orcgpu_host;

static int expand(const std::vector<uint8_t>& raw, int compression, uint64_t block_size, std::vector<uint8_t>& out) {
  if (compression == ORCGPU_COMP_NONE) {
    out = raw;
    return ORCGPU_OK;
  }
  if (compression < 0 || compression > 5) return ORCGPU_OUT_OF_SPEC;
  uint8_t* p = nullptr;
  size_t n = 0;
  const int rc = oo_stream_decompress(raw.data(), raw.size(), compression, block_size ? (size_t)std::min<uint64_t>(block_size, 1u << 26) : 262144, &p, &n);
  if (rc == 0) out.assign(p, p + n);
  if (p) oo_free(p);
  return rc ? ORCGPU_BUILD_DECODER : ORCGPU_OK;
}

struct Summary {
  uint64_t rows = 0, stripes = 0, types = 0, streams = 0, index_entries = 0, splits_ok = 0;
};

static int walk(const uint8_t* data, uint64_t len, Summary& S) {
  BytesChunkReader r(data, len);
  FileMetadata md;
  MetaErr e;
  const uint64_t file_len = r.len();
  if (file_len == 0) return ORCGPU_OUT_OF_SPEC;
  const uint64_t tail_len = std::min<uint64_t>(file_len, 16 * 1024);
  std::vector<uint8_t> tail;
  if (!r.get_bytes(file_len - tail_len, tail_len, tail)) return ORCGPU_IO_ERROR;
  uint64_t ps_len = 0, footer_length = 0;
  int rc = parse_postscript(tail.data(), tail.size(), md, ps_len, footer_length, e);
  if (rc) return rc;
  const uint64_t footer_end = file_len - 1 - ps_len;
  if (footer_length > footer_end) return ORCGPU_OUT_OF_SPEC;
  std::vector<uint8_t> raw_footer, footer;
  if (!r.get_bytes(footer_end - footer_length, footer_length, raw_footer)) return ORCGPU_IO_ERROR;
  if ((rc = expand(raw_footer, md.compression, md.block_size, footer))) return rc;
  if ((rc = parse_footer(footer.data(), footer.size(), md, e))) return rc;
  S.rows = md.number_of_rows;
  S.stripes = md.stripes.size();
  S.types = md.types.size();
  // the column shard's weights and deal over the root columns (orcgpu_reader_set_shard)
  if (!md.types.empty()) {
    std::vector<double> w;
    for (uint32_t k : md.types[0].subtypes) w.push_back(type_weight(md.types, k));
    std::vector<uint32_t> rank_of(w.size());
    if (!w.empty()) lpt_deal(w.data(), (uint32_t)w.size(), 8, rank_of.data());
  }
  for (const StripeInfo& si : md.stripes) {
    std::vector<uint8_t> raw, plain;
    if ((rc = stripe_footer_raw(r, si, raw, e))) return rc;
    if ((rc = expand(raw, md.compression, md.block_size, plain))) return rc;
    StripeFooter sf;
    if ((rc = parse_stripe_footer(si, plain, sf, e))) return rc;
    S.streams += sf.streams.size();
    // every column's ROW_INDEX (+ Bloom filters, statistics): read_stripe_index with want_stats
    std::vector<uint32_t> columns;
    for (uint32_t c = 0; c < md.types.size(); c++) columns.push_back(c);
    std::vector<std::vector<uint8_t>> raws, plains;
    IndexSections is;
    if ((rc = index_gather(r, sf, columns, true, raws, is))) return rc;
    plains.resize(raws.size());
    bool ok = true;
    for (size_t k = 0; k < raws.size() && ok; k++) ok = expand(raws[k], md.compression, md.block_size, plains[k]) == ORCGPU_OK;
    if (!ok) continue;  // (the reader then decodes the stripe whole)
    if (index_parse(sf, is, plains, true) != ORCGPU_OK) continue;
    for (const ColumnIndex& ci : sf.index) {
      S.index_entries += ci.n_groups;
      if (!ci.per_group || ci.column >= md.types.size()) continue;
      const int kind = md.types[ci.column].kind;
      const int enc = ci.column < sf.encodings.size() ? sf.encodings[ci.column].first : 0;
      bool has_present = false;
      for (auto& s : sf.streams) has_present = has_present || (s.column == ci.column && s.kind == ORCGPU_S_PRESENT);
      for (uint32_t g = 0; g < ci.n_groups; g++) {
        EntrySplit es;
        S.splits_ok += split_index_entry(kind, enc, has_present, md.compression != ORCGPU_COMP_NONE, ci.positions.data() + (size_t)g * ci.per_group, ci.per_group, es);
      }
    }
  }
  return ORCGPU_OK;
}

static std::vector<uint8_t> slurp(const char* path) {
  std::vector<uint8_t> b;
  FILE* f = fopen(path, "rb");
  if (!f) return b;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  b.resize(n > 0 ? (size_t)n : 0);
  if (n > 0 && fread(b.data(), 1, (size_t)n, f) != (size_t)n) b.clear();
  fclose(f);
  return b;
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string cmd = argv[1];
  if (cmd == "walk" && argc >= 3) {
    const std::vector<uint8_t> b = slurp(argv[2]);
    Summary S;
    const int rc = walk(b.data(), b.size(), S);
    printf("{\"status\": %d, \"rows\": %" PRIu64 ", \"stripes\": %" PRIu64 ", \"types\": %" PRIu64 ", \"streams\": %" PRIu64 ", \"index_entries\": %" PRIu64 ", \"splits_ok\": %" PRIu64 "}\n",
           rc, S.rows, S.stripes, S.types, S.streams, S.index_entries, S.splits_ok);
    return 0;
  }
  if (cmd == "fuzz" && argc >= 5) {
    const std::vector<uint8_t> clean = slurp(argv[2]);
    std::mt19937_64 rng(strtoull(argv[3], nullptr, 10));
    const int n = atoi(argv[4]);
    std::map<int, int> outcomes;
    Summary S0;
    const int rc0 = walk(clean.data(), clean.size(), S0);
    // where the stripe footers lie (of the clean file): half of the hits go there
    std::vector<std::pair<uint64_t, uint64_t>> footers;
    {
      BytesChunkReader r(clean.data(), clean.size());
      FileMetadata md;
      MetaErr e;
      uint64_t ps = 0, fl = 0;
      const uint64_t tl = std::min<uint64_t>(clean.size(), 16384);
      if (!clean.empty() && parse_postscript(clean.data() + clean.size() - tl, tl, md, ps, fl, e) == 0 && fl <= clean.size() - 1 - ps) {
        std::vector<uint8_t> raw(clean.begin() + (clean.size() - 1 - ps - fl), clean.begin() + (clean.size() - 1 - ps)), plain;
        if (expand(raw, md.compression, md.block_size, plain) == 0 && parse_footer(plain.data(), plain.size(), md, e) == 0)
          for (auto& si : md.stripes)
            if (si.footer_length && si.offset + si.index_length + si.data_length + si.footer_length <= clean.size())
              footers.push_back({si.offset + si.index_length + si.data_length, si.footer_length});
      }
    }
    for (int k = 0; k < n && !clean.empty(); k++) {
      std::vector<uint8_t> m = clean;
      const int what = (int)(rng() % 8);
      if (what == 0) {
        m.resize((size_t)(rng() % clean.size()));  // a truncation anywhere
      } else if (what == 1) {
        m.resize(clean.size() - 1 - (size_t)(rng() % std::min<size_t>(clean.size() - 1 ? clean.size() - 1 : 1, 64)));  // ... near the end
      } else {
        const int hits = 1 + (int)(rng() % 4);
        for (int h = 0; h < hits; h++) {
          uint64_t lo = clean.size() > 16384 ? clean.size() - 16384 : 0, span = clean.size() - lo;
          if (!footers.empty() && (rng() & 1)) {
            auto& f = footers[rng() % footers.size()];
            lo = f.first;
            span = f.second;
          }
          const size_t at = (size_t)(lo + rng() % span);
          if (what < 6) m[at] ^= (uint8_t)(1u << (rng() % 8));
          else m[at] = (uint8_t)rng();
        }
      }
      Summary S;
      outcomes[walk(m.data(), m.size(), S)]++;
    }
    printf("{\"clean_status\": %d, \"clean_rows\": %" PRIu64 ", \"outcomes\": {", rc0, S0.rows);
    bool first = true;
    for (auto& kv : outcomes) {
      printf("%s\"%d\": %d", first ? "" : ", ", kv.first, kv.second);
      first = false;
    }
    printf("}}\n");
    return 0;
  }
  if (cmd == "tz" && argc >= 3) {
    TzTable T;
    const bool ok = load_tzif(argv[2], T);
    printf("{\"ok\": %d, \"transitions\": %zu, \"utc\": %d}\n", ok ? 1 : 0, T.at.size(), T.utc ? 1 : 0);
    if (ok) (void)tz_orc_epoch(T);
    return 0;
  }
  if (cmd == "select" && argc >= 4) {
    std::mt19937_64 rng(strtoull(argv[2], nullptr, 10));
    const int n = atoi(argv[3]);
    uint64_t checked = 0;
    for (int k = 0; k < n; k++) {
      std::vector<orcgpu_row_selector> raw((size_t)(rng() % 12));
      for (auto& s : raw) {
        s.row_count = rng() % 5000;
        s.skip = (int32_t)(rng() & 1);
      }
      std::vector<RowSel> sel = sel_normalise(raw.data(), (uint32_t)raw.size());
      const uint64_t total = sel_row_count(sel);
      const uint64_t rows = 1 + rng() % 20000, batch = 1 + rng() % 3000;
      std::vector<RowSel> rest = sel;
      std::vector<RowSel> mine = sel_split_off(rest, rows);
      if (sel_row_count(mine) + sel_row_count(rest) != total) abort();
      const std::vector<SelBatch> b = sel_batches(mine, rows, batch);
      uint64_t prev_end = 0;
      for (auto& sb : b) {
        if (sb.len == 0 || sb.len > batch || sb.start < prev_end || sb.start + sb.len > rows) abort();
        prev_end = sb.start + sb.len;
      }
      (void)sel_intersect(mine, sel, rows);
      checked += b.size();
    }
    printf("{\"selections\": %d, \"batches\": %" PRIu64 "}\n", n, checked);
    return 0;
  }
  return 2;
}
