/*
 * tpchgen.c -- TPC-H-shaped `lineitem` column generator (host side, plain C, seeded).
 *
 * BASELINE.json's metric is quoted on a TPC-H lineitem stripe (schema: the reference's
 * scripts/convert_tpch.py:46-63).  dbgen is not available (scripts/generate-tpch.sh:29-31 needs
 * docker + network), so this file follows the column domains of the TPC-H specification 4.2.3
 * instead: sparse order keys (8 of every 32 used), 1..7 lines per order, part / supplier keys,
 * quantity 1..50, extended price = quantity x retail price of the part, discount 0.00..0.10,
 * tax 0.00..0.08, return flag / line status from the dates against 1995-06-17, ship / commit /
 * receipt dates from the order date, four ship instructions, seven ship modes, and comments cut
 * out of a pool of text produced by the specification's sentence grammar (dbgen does the same
 * with a 300 MB pool; this one is 32 MiB).  "TPC-H-shaped", not TPC-H: the random streams are not
 * dbgen's.  Not on the decode path; used by bench.py, the tests and profiles/ to make inputs.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline uint64_t sm64(uint64_t* s) {
  uint64_t z = (*s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
static inline uint64_t rnd(uint64_t* s, uint64_t lo, uint64_t hi) { return lo + sm64(s) % (hi - lo + 1); }

static const char* NOUNS[] = {"foxes", "ideas", "theodolites", "pinto beans", "instructions", "dependencies", "excuses", "platelets",
                              "asymptotes", "courts", "dolphins", "multipliers", "sauternes", "warthogs", "frets", "dinos",
                              "attainments", "somas", "Tiresias'", "patterns", "forges", "braids", "hockey players", "frays",
                              "warhorses", "dugouts", "notornis", "epitaphs", "pearls", "tithes", "waters", "orbits", "gifts",
                              "sheaves", "depths", "sentiments", "decoys", "realms", "pains", "grouches", "escapades"};
static const char* VERBS[] = {"sleep", "wake", "are", "cajole", "haggle", "nag", "use", "boost", "affix", "detect", "integrate",
                              "maintain", "nod", "was", "lose", "sublate", "solve", "thrash", "promise", "engage", "hinder",
                              "print", "x-ray", "breach", "eat", "grow", "impress", "mold", "poach", "serve", "run", "dazzle",
                              "snooze", "doze", "unwind", "kindle", "play", "hang", "believe", "doubt"};
static const char* ADJS[] = {"furious", "sly", "careful", "blithe", "quick", "fluffy", "slow", "quiet", "ruthless", "thin", "close",
                             "dogged", "daring", "brave", "stealthy", "permanent", "enticing", "idle", "busy", "regular", "final",
                             "ironic", "even", "bold", "silent"};
static const char* ADVS[] = {"sometimes", "always", "never", "furiously", "slyly", "carefully", "blithely", "quickly", "fluffily",
                             "slowly", "quietly", "ruthlessly", "thinly", "closely", "doggedly", "daringly", "bravely",
                             "stealthily", "permanently", "enticingly", "idly", "busily", "regularly", "finally", "ironically",
                             "evenly", "boldly", "silently"};
static const char* PREPS[] = {"about", "above", "according to", "across", "after", "against", "along", "alongside of", "among",
                              "around", "at", "atop", "before", "behind", "beneath", "beside", "besides", "between", "beyond", "by",
                              "despite", "during", "except", "for", "from", "in place of", "inside", "instead of", "into", "near",
                              "of", "on", "outside", "over", "past", "since", "through", "throughout", "to", "toward", "under",
                              "until", "up", "upon", "without", "with", "within"};
static const char* AUXS[] = {"do", "may", "might", "shall", "will", "would", "can", "could", "should", "ought to", "must",
                             "will have to", "shall have to", "could have to", "should have to", "must have to", "need to", "try to"};
static const char* TERMS[] = {".", ";", ":", "?", "!", "--"};
#define NEL(a) (sizeof(a) / sizeof((a)[0]))

typedef struct {
  char* p;
  size_t len, cap;
  uint64_t rs;
} pool_t;
static void put(pool_t* t, const char* s) {
  size_t n = strlen(s);
  if (t->len + n + 1 > t->cap) n = t->cap > t->len ? t->cap - t->len : 0;
  memcpy(t->p + t->len, s, n);
  t->len += n;
}
#define PICK(arr) arr[sm64(&t->rs) % NEL(arr)]
static void noun_phrase(pool_t* t) {
  switch (sm64(&t->rs) % 4) {
    case 0: put(t, PICK(NOUNS)); break;
    case 1: put(t, PICK(ADJS)); put(t, " "); put(t, PICK(NOUNS)); break;
    case 2: put(t, PICK(ADJS)); put(t, ", "); put(t, PICK(ADJS)); put(t, " "); put(t, PICK(NOUNS)); break;
    default: put(t, PICK(ADVS)); put(t, " "); put(t, PICK(ADJS)); put(t, " "); put(t, PICK(NOUNS)); break;
  }
}
static void verb_phrase(pool_t* t) {
  switch (sm64(&t->rs) % 4) {
    case 0: put(t, PICK(VERBS)); break;
    case 1: put(t, PICK(AUXS)); put(t, " "); put(t, PICK(VERBS)); break;
    case 2: put(t, PICK(VERBS)); put(t, " "); put(t, PICK(ADVS)); break;
    default: put(t, PICK(AUXS)); put(t, " "); put(t, PICK(VERBS)); put(t, " "); put(t, PICK(ADVS)); break;
  }
}
static void prep_phrase(pool_t* t) {
  put(t, PICK(PREPS));
  put(t, " the ");
  noun_phrase(t);
}
static void sentence(pool_t* t) {
  switch (sm64(&t->rs) % 5) {
    case 0: noun_phrase(t); put(t, " "); verb_phrase(t); break;
    case 1: noun_phrase(t); put(t, " "); verb_phrase(t); put(t, " "); prep_phrase(t); break;
    case 2: noun_phrase(t); put(t, " "); verb_phrase(t); put(t, " "); noun_phrase(t); break;
    case 3: noun_phrase(t); put(t, " "); prep_phrase(t); put(t, " "); verb_phrase(t); break;
    default: noun_phrase(t); put(t, " "); prep_phrase(t); put(t, " "); verb_phrase(t); put(t, " "); prep_phrase(t); break;
  }
  put(t, PICK(TERMS));
  put(t, " ");
}

/* days since 1970-01-01 of 1992-01-01 and 1998-12-31; the "current date" of the specification (1995-06-17) */
#define D_START 8035
#define D_END 10591
#define D_CURRENT 9298

typedef struct {
  int64_t *orderkey, *partkey, *suppkey, *quantity, *extendedprice, *discount, *tax; /* decimals: unscaled (x100) */
  int32_t *linenumber, *shipdate, *commitdate, *receiptdate, *comment_len;
  uint8_t *returnflag, *linestatus, *shipinstruct, *shipmode; /* dictionary indexes (sorted dictionaries) */
  uint8_t* comment; /* concatenated comment bytes, capacity 44 * n_rows */
} lineitem_cols;

/* Fills n_rows rows (any NULL column is skipped); returns the number of comment bytes written.
 * Dictionaries (sorted, as ORC writers store them): returnflag A N R; linestatus F O;
 * shipinstruct COLLECT COD / DELIVER IN PERSON / NONE / TAKE BACK RETURN;
 * shipmode AIR FOB MAIL RAIL REG AIR SHIP TRUCK. */
uint64_t orcgen_lineitem(uint64_t seed, uint64_t n_rows, uint64_t scale_factor, lineitem_cols* c) {
  const uint64_t pool_bytes = 32ull << 20;
  pool_t pool = {(char*)malloc(pool_bytes + 64), 0, pool_bytes, seed ^ 0x7e57ull};
  while (pool.len + 400 < pool.cap) sentence(&pool);
  uint64_t rs = seed;
  const uint64_t n_parts = 200000ull * scale_factor, n_supp = 10000ull * scale_factor;
  uint64_t row = 0, order = 0, cbytes = 0;
  while (row < n_rows) {
    const int64_t okey = (int64_t)((order / 8) * 32 + order % 8 + 1);
    order++;
    const int lines = (int)rnd(&rs, 1, 7);
    const int32_t odate = (int32_t)rnd(&rs, D_START, D_END - 151);
    for (int l = 0; l < lines && row < n_rows; l++, row++) {
      const uint64_t pk = rnd(&rs, 1, n_parts);
      const uint64_t si = rnd(&rs, 0, 3);
      const uint64_t sk = (pk + si * (n_supp / 4 + (pk - 1) / n_supp)) % n_supp + 1;
      const int64_t q = (int64_t)rnd(&rs, 1, 50);
      const int64_t retail = 90000 + (int64_t)((pk / 10) % 20001) + 100 * (int64_t)(pk % 1000); /* cents */
      const int32_t sdate = odate + (int32_t)rnd(&rs, 1, 121);
      const int32_t cdate = odate + (int32_t)rnd(&rs, 30, 90);
      const int32_t rdate = sdate + (int32_t)rnd(&rs, 1, 30);
      const uint64_t disc = rnd(&rs, 0, 10), tax = rnd(&rs, 0, 8);
      const uint64_t ra = sm64(&rs) & 1, instr = rnd(&rs, 0, 3), mode = rnd(&rs, 0, 6);
      const uint64_t clen = rnd(&rs, 10, 43), coff = sm64(&rs) % (pool.len - 64);
      if (c->orderkey) c->orderkey[row] = okey;
      if (c->partkey) c->partkey[row] = (int64_t)pk;
      if (c->suppkey) c->suppkey[row] = (int64_t)sk;
      if (c->linenumber) c->linenumber[row] = l + 1;
      if (c->quantity) c->quantity[row] = q * 100;
      if (c->extendedprice) c->extendedprice[row] = q * retail;
      if (c->discount) c->discount[row] = (int64_t)disc;
      if (c->tax) c->tax[row] = (int64_t)tax;
      if (c->returnflag) c->returnflag[row] = rdate <= D_CURRENT ? (ra ? 2 : 0) : 1; /* R / A, else N */
      if (c->linestatus) c->linestatus[row] = sdate > D_CURRENT ? 1 : 0;            /* O, else F   */
      if (c->shipdate) c->shipdate[row] = sdate;
      if (c->commitdate) c->commitdate[row] = cdate;
      if (c->receiptdate) c->receiptdate[row] = rdate;
      if (c->shipinstruct) c->shipinstruct[row] = (uint8_t)instr;
      if (c->shipmode) c->shipmode[row] = (uint8_t)mode;
      if (c->comment_len) c->comment_len[row] = (int32_t)clen;
      if (c->comment) {
        memcpy(c->comment + cbytes, pool.p + coff, clen);
        cbytes += clen;
      }
    }
  }
  free(pool.p);
  return cbytes;
}

/* zigzag varints of 64-bit values (Decimal DATA of precision <= 18; encoding/decimal.rs:28-52 reads them as i128) */
int orcgen_varint64(const int64_t* v, size_t n, uint8_t** out, size_t* out_len) {
  uint8_t* p = (uint8_t*)malloc(n * 10 + 64);
  size_t o = 0;
  for (size_t i = 0; i < n; i++) {
    uint64_t u = ((uint64_t)v[i] << 1) ^ (uint64_t)(v[i] >> 63);
    while (u >= 0x80) {
      p[o++] = (uint8_t)(u | 0x80);
      u >>= 7;
    }
    p[o++] = (uint8_t)u;
  }
  *out = p;
  *out_len = o;
  return 0;
}
