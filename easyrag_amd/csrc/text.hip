// Text side of the BM25 index build, host only (second half): the reference tokenises every
// node with jieba and feeds Python token lists to rank_bm25 / bm25s
// (/root/reference/src/easyrag/custom/retrievers.py:72-76, 94-118; src/easyrag/pipeline/pipeline.py:176-178).  Here:
//   erh_vocab_*    token bytes -> term ids through an open-addressing hash table; ids follow first appearance, exactly
//                  like the Python dict loop they replace (easyrag_amd/index.py: vocab_ids), so the CSR built from them
//                  is bit-identical;
//   erh_cutter_*   a dictionary cutter with jieba's sentence-splitting rules and its DAG / maximum-log-probability route
//                  (jieba 0.42.1, Tokenizer.cut(sentence, cut_all=False)) over a caller-supplied dictionary in jieba's
//                  "word freq [tag]" text format.  jieba's default call (HMM=True) also runs an HMM over runs of
//                  out-of-dictionary characters (jieba/finalseg); the algorithm is here (viterbi over B M E S, finalseg's
//                  own block splitting), its trained tables ship with jieba and are supplied by the caller as text
//                  (erh_cutter_set_hmm; INTEGRATION.md shows the dump).  Without them the cutter is HMM=False.
// No device code in this file; it is part of libeasyrag_hip.so so that one library serves the whole retriever shim.
#include "../../include/easyrag_hip.h"

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

// FNV-1a, 64 bit, byte by byte (so a fragment's hash extends to the next character in O(1)), then a finaliser
constexpr uint64_t kFnvBasis = 1469598103934665603ull;
inline uint64_t fnv_extend(uint64_t h, const char *p, size_t n) {
    for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)p[i]; h *= 1099511628211ull; }
    return h;
}
inline uint64_t fnv_final(uint64_t h) {
    h ^= h >> 32;
    h *= 0x9e3779b97f4a7c15ull;
    h ^= h >> 29;
    return h;
}
inline uint64_t hash_bytes(const char *p, size_t n) { return fnv_final(fnv_extend(kFnvBasis, p, n)); }

}  // namespace

struct erh_vocab {
    std::vector<char> arena;                 // token bytes, back to back
    std::vector<int64_t> tok_off;            // token id -> [tok_off[id], tok_off[id + 1])
    std::vector<int32_t> slots;              // open addressing: -1 empty, else token id
    std::vector<uint64_t> hashes;            // per token id
    size_t mask = 0;

    erh_vocab() { tok_off.push_back(0); slots.assign(1024, -1); mask = 1023; }

    void grow() {
        const size_t n = slots.size() * 2;
        std::vector<int32_t> s(n, -1);
        const size_t m = n - 1;
        for (size_t id = 0; id < hashes.size(); ++id) {
            size_t i = hashes[id] & m;
            while (s[i] >= 0) i = (i + 1) & m;
            s[i] = (int32_t)id;
        }
        slots.swap(s);
        mask = m;
    }

    int32_t find_or_add(const char *p, size_t n, bool add) { return find_or_add(hash_bytes(p, n), p, n, add); }

    int32_t find_or_add(uint64_t h, const char *p, size_t n, bool add) {
        size_t i = h & mask;
        for (;;) {
            const int32_t id = slots[i];
            if (id < 0) break;
            if (hashes[id] == h) {
                const int64_t b = tok_off[id], e = tok_off[id + 1];
                if ((size_t)(e - b) == n && memcmp(arena.data() + b, p, n) == 0) return id;
            }
            i = (i + 1) & mask;
        }
        if (!add) return -1;
        if (hashes.size() >= 0x7ffffffeull) return -2;
        const int32_t id = (int32_t)hashes.size();
        arena.insert(arena.end(), p, p + n);
        tok_off.push_back((int64_t)arena.size());
        hashes.push_back(h);
        slots[i] = id;
        if (hashes.size() * 10 > slots.size() * 6) grow();       // load factor <= 0.6
        return id;
    }
};

constexpr double kHmmMin = -3.14e100;       // jieba.finalseg.MIN_FLOAT
struct erh_hmm_emit { double p[4]; };        // states in the order B, E, M, S (alphabetical: ties go to the later letter, as Python's tuple max does)
struct erh_cutter {
    erh_vocab words;                         // jieba's FREQ keys: every word and every prefix of a word ...
    std::vector<int64_t> freq;               // ... and their frequencies by id (prefixes that are not words: 0)
    double total = 0.0;
    // jieba.finalseg: the HMM that regroups runs of out-of-dictionary single characters (cut(..., HMM=True), jieba's default)
    bool has_hmm = false;
    double hmm_start[4] = {kHmmMin, kHmmMin, kHmmMin, kHmmMin};
    double hmm_trans[4][4];
    std::unordered_map<uint32_t, erh_hmm_emit> hmm_emit;

    void set(const char *p, size_t n, int64_t f, bool overwrite) {
        const int32_t id = words.find_or_add(p, n, true);
        if ((size_t)id >= freq.size()) { freq.resize((size_t)id + 1, 0); freq[id] = f; }
        else if (overwrite) freq[id] = f;
    }
};

namespace {

// UTF-8 -> code points + byte offset of each (offsets has one more entry: the end).  Malformed bytes decode as
// themselves (one "character" per byte, value 0xDC80 + byte, Python's surrogateescape convention), so cutting never fails.
// The three-byte form of a surrogate code point (what str.encode("utf-8", "surrogatepass") writes for a lone surrogate in a
// Python str) is ONE character, as it is for jieba, which walks the str.
void decode_utf8(const char *s, int64_t n, std::vector<uint32_t> &cp, std::vector<int64_t> &off) {
    cp.clear();
    off.clear();
    int64_t i = 0;
    while (i < n) {
        const unsigned char c = (unsigned char)s[i];
        uint32_t v = 0xDC80u + c;
        int len = 1;
        if (c < 0x80) { v = c; }
        else if ((c & 0xE0) == 0xC0 && i + 1 < n && ((unsigned char)s[i + 1] & 0xC0) == 0x80) {
            const uint32_t w = ((uint32_t)(c & 0x1F) << 6) | ((unsigned char)s[i + 1] & 0x3F);
            if (w >= 0x80) { v = w; len = 2; }
        } else if ((c & 0xF0) == 0xE0 && i + 2 < n && ((unsigned char)s[i + 1] & 0xC0) == 0x80 &&
                   ((unsigned char)s[i + 2] & 0xC0) == 0x80) {
            const uint32_t w = ((uint32_t)(c & 0x0F) << 12) | (((uint32_t)(unsigned char)s[i + 1] & 0x3F) << 6) |
                               ((unsigned char)s[i + 2] & 0x3F);
            if (w >= 0x800) { v = w; len = 3; }
        } else if ((c & 0xF8) == 0xF0 && i + 3 < n && ((unsigned char)s[i + 1] & 0xC0) == 0x80 &&
                   ((unsigned char)s[i + 2] & 0xC0) == 0x80 && ((unsigned char)s[i + 3] & 0xC0) == 0x80) {
            const uint32_t w = ((uint32_t)(c & 0x07) << 18) | (((uint32_t)(unsigned char)s[i + 1] & 0x3F) << 12) |
                               (((uint32_t)(unsigned char)s[i + 2] & 0x3F) << 6) | ((unsigned char)s[i + 3] & 0x3F);
            if (w >= 0x10000 && w <= 0x10FFFF) { v = w; len = 4; }
        }
        cp.push_back(v);
        off.push_back(i);
        i += len;
    }
    off.push_back(n);
}

// jieba.re_han_default: [一-鿕a-zA-Z0-9+#&\._%\-]
inline bool is_han_class(uint32_t c) {
    if (c >= 0x4E00 && c <= 0x9FD5) return true;
    if ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9')) return true;
    return c == '+' || c == '#' || c == '&' || c == '.' || c == '_' || c == '%' || c == '-';
}
// Python's \s on str patterns (= str.isspace)
inline bool is_space(uint32_t c) {
    if ((c >= 0x09 && c <= 0x0D) || (c >= 0x1C && c <= 0x20)) return true;
    switch (c) {
        case 0x85: case 0xA0: case 0x1680: case 0x2028: case 0x2029: case 0x202F: case 0x205F: case 0x3000: return true;
        default: return c >= 0x2000 && c <= 0x200A;
    }
}
inline bool is_eng(uint32_t c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9'); }

// one block of han-class characters [b, e): jieba's get_DAG + calc + __cut_DAG_NO_HMM; appends token END positions
// (character indices) to `ends`
inline bool is_pure_han(uint32_t c) { return c >= 0x4E00 && c <= 0x9FD5; }   // finalseg.re_han: [\u4E00-\u9FD5]

// jieba.finalseg.viterbi + __cut over the pure-han run cp[b, e): appends token END positions
void hmm_cut_han(const erh_cutter *c, const std::vector<uint32_t> &cp, int64_t b, int64_t e, std::vector<int64_t> &ends) {
    const int64_t n = e - b;
    static const int prev[4][2] = {{1, 3}, {0, 2}, {2, 0}, {3, 1}};      // PrevStatus: B <- E S, E <- B M, M <- M B, S <- S E
    std::vector<double> V((size_t)n * 4);
    std::vector<int8_t> from((size_t)n * 4, -1);
    auto emit = [&](uint32_t ch, int y) -> double {
        const auto it = c->hmm_emit.find(ch);
        return it == c->hmm_emit.end() ? kHmmMin : it->second.p[y];
    };
    for (int y = 0; y < 4; ++y) V[(size_t)y] = c->hmm_start[y] + emit(cp[b], y);
    for (int64_t t = 1; t < n; ++t) {
        for (int y = 0; y < 4; ++y) {
            const double em = emit(cp[b + t], y);
            double best = 0.0;
            int bs = -1;
            for (int j = 0; j < 2; ++j) {                                // max over (prob, state): the later letter wins a tie
                const int y0 = prev[y][j];
                const double v = V[(size_t)(t - 1) * 4 + (size_t)y0] + c->hmm_trans[y0][y] + em;
                if (bs < 0 || v > best || (v == best && y0 > bs)) { best = v; bs = y0; }
            }
            V[(size_t)t * 4 + (size_t)y] = best;
            from[(size_t)t * 4 + (size_t)y] = (int8_t)bs;
        }
    }
    const double ve = V[(size_t)(n - 1) * 4 + 1], vs = V[(size_t)(n - 1) * 4 + 3];
    int st = (vs >= ve) ? 3 : 1;                                         // max((V[E], 'E'), (V[S], 'S'))
    std::vector<int8_t> path((size_t)n);
    for (int64_t t = n - 1; t >= 0; --t) { path[(size_t)t] = (int8_t)st; if (t > 0) st = from[(size_t)t * 4 + (size_t)st]; }
    int64_t nexti = 0;
    for (int64_t i = 0; i < n; ++i) {                                    // 'B' opens, 'E' closes [begin, i], 'S' stands alone
        const int y = path[(size_t)i];
        if (y == 1 || y == 3) { ends.push_back(b + i + 1); nexti = i + 1; }
    }
    if (nexti < n) ends.push_back(b + n);
}

// jieba.finalseg.cut over cp[b, e) (characters of the han class): pure-han runs through the HMM, the rest split by
// re_skip = ([a-zA-Z0-9]+(?:\.\d+)?%?) with the text between two matches kept whole
void hmm_cut(const erh_cutter *c, const std::vector<uint32_t> &cp, int64_t b, int64_t e, std::vector<int64_t> &ends) {
    int64_t i = b;
    while (i < e) {
        if (is_pure_han(cp[i])) {
            int64_t j = i;
            while (j < e && is_pure_han(cp[j])) ++j;
            hmm_cut_han(c, cp, i, j, ends);
            i = j;
            continue;
        }
        int64_t stop = i;
        while (stop < e && !is_pure_han(cp[stop])) ++stop;                 // the non-han block [i, stop)
        int64_t run = i;                                                  // start of the pending unmatched text
        while (i < stop) {
            if (is_eng(cp[i])) {
                if (run < i) ends.push_back(i);                           // text before the match
                int64_t j = i;
                while (j < stop && is_eng(cp[j])) ++j;
                if (j + 1 < stop && cp[j] == '.' && cp[j + 1] >= '0' && cp[j + 1] <= '9') {
                    ++j;
                    while (j < stop && cp[j] >= '0' && cp[j] <= '9') ++j;
                }
                if (j < stop && cp[j] == '%') ++j;
                ends.push_back(j);
                i = run = j;
            } else {
                ++i;
            }
        }
        if (run < stop) ends.push_back(stop);
    }
}

void cut_block(const erh_cutter *c, const char *s, const std::vector<int64_t> &off, const std::vector<uint32_t> &cp, int64_t b,
               int64_t e, std::vector<int64_t> &ends, bool hmm) {
    const int64_t N = e - b;
    // DAG edges k -> i (sentence[k : i + 1] is a word) with the word's frequency, found by extending the fragment's
    // hash one character at a time; dag_off[k] .. dag_off[k + 1] index the edge arrays
    std::vector<int32_t> dag_off((size_t)N + 1, 0), dag_to;
    std::vector<int64_t> dag_f;
    for (int64_t k = 0; k < N; ++k) {
        uint64_t h = kFnvBasis;
        const char *p0 = s + off[b + k];
        bool any = false;
        for (int64_t i = k; i < N; ++i) {                                // frag = sentence[k : i + 1]
            h = fnv_extend(h, s + off[b + i], (size_t)(off[b + i + 1] - off[b + i]));
            const int32_t id = const_cast<erh_vocab &>(c->words).find_or_add(fnv_final(h), p0, (size_t)(off[b + i + 1] - off[b + k]), false);
            if (id < 0) break;                                           // `while i < N and frag in self.FREQ`
            if (c->freq[id]) { dag_to.push_back((int32_t)i); dag_f.push_back(c->freq[id]); any = true; }
        }
        if (!any) { dag_to.push_back((int32_t)k); dag_f.push_back(1); }   // a character on its own: `FREQ.get(...) or 1`
        dag_off[k + 1] = (int32_t)dag_to.size();
    }
    const double logtotal = std::log(c->total);
    std::vector<double> rp((size_t)N + 1, 0.0);
    std::vector<int32_t> rx((size_t)N + 1, 0);
    for (int64_t idx = N - 1; idx >= 0; --idx) {
        bool first = true;
        double best = 0.0;
        int32_t bx = 0;
        for (int32_t ed = dag_off[idx]; ed < dag_off[idx + 1]; ++ed) {   // max over tuples (log-probability, x)
            const int32_t x = dag_to[ed];
            const double v = std::log((double)dag_f[ed]) - logtotal + rp[x + 1];
            if (first || v > best || (v == best && x > bx)) { best = v; bx = x; first = false; }
        }
        rp[idx] = best;
        rx[idx] = bx;
    }
    if (hmm) {
        // __cut_DAG: single characters collect in buf; when a longer word (or the end) arrives, a one-character buf is a
        // token, a buf that is itself a dictionary word goes out character by character, anything else through the HMM
        int64_t x = 0, bs = -1;                                          // buf = characters [bs, x)
        auto flush = [&](int64_t be) {
            if (bs < 0) return;
            if (be - bs == 1) {
                ends.push_back(b + be);
            } else {
                const int32_t id = const_cast<erh_vocab &>(c->words).find_or_add(s + off[b + bs], (size_t)(off[b + be] - off[b + bs]), false);
                if (id >= 0 && c->freq[id]) { for (int64_t q = bs; q < be; ++q) ends.push_back(b + q + 1); }
                else hmm_cut(c, cp, b + bs, b + be, ends);
            }
            bs = -1;
        };
        while (x < N) {
            const int64_t y = rx[x] + 1;
            if (y - x == 1) {
                if (bs < 0) bs = x;
            } else {
                flush(x);
                ends.push_back(b + y);
            }
            x = y;
        }
        flush(N);
        return;
    }
    int64_t x = 0;
    bool buf = false;                                                    // a run of single ASCII letters / digits is open
    while (x < N) {
        const int64_t y = rx[x] + 1;
        if (y - x == 1 && is_eng(cp[b + x])) {
            buf = true;
        } else {
            if (buf) { ends.push_back(b + x); buf = false; }
            ends.push_back(b + y);
        }
        x = y;
    }
    if (buf) ends.push_back(b + N);
}

// whole sentence: jieba's cut() block splitting around cut_block; token END positions in characters
void cut_text(const erh_cutter *c, const char *text, int64_t n_bytes, std::vector<uint32_t> &cp, std::vector<int64_t> &off,
              std::vector<int64_t> &ends, bool hmm) {
    ends.clear();
    decode_utf8(text, n_bytes, cp, off);
    const int64_t n = (int64_t)cp.size();
    int64_t i = 0;
    while (i < n) {
        if (is_han_class(cp[i])) {                                       // re_han.split: maximal runs of the han class
            int64_t e = i;
            while (e < n && is_han_class(cp[e])) ++e;
            cut_block(c, text, off, cp, i, e, ends, hmm);
            i = e;
        } else if (cp[i] == '\r' && i + 1 < n && cp[i + 1] == '\n') {   // re_skip: (\r\n|\s), the pair first
            ends.push_back(i + 2);
            i += 2;
        } else {                                                         // a white-space character, or any other one, alone
            ends.push_back(i + 1);
            ++i;
        }
    }
}

}  // namespace

extern "C" {

int erh_vocab_create(erh_vocab **out) {
    if (!out) return ERH_ERR_INVALID;
    *out = new (std::nothrow) erh_vocab();
    return *out ? ERH_OK : ERH_ERR_NOMEM;
}

int erh_vocab_destroy(erh_vocab *v) {
    if (!v) return ERH_ERR_INVALID;
    delete v;
    return ERH_OK;
}

int64_t erh_vocab_size(const erh_vocab *v) { return v ? (int64_t)v->hashes.size() : -1; }

int erh_vocab_encode(erh_vocab *v, const char *bytes, const int64_t *doc_off, int64_t n_docs, int sep, int add,
                     int32_t *out_ids, int64_t cap, int32_t *out_lens, int64_t *n_out) {
    if (!v || !doc_off || n_docs < 0 || !n_out || (n_docs > 0 && !out_lens) || sep < 0 || sep > 255) return ERH_ERR_INVALID;
    int64_t w = 0;
    try {
        for (int64_t d = 0; d < n_docs; ++d) {
            const int64_t b = doc_off[d], e = doc_off[d + 1];
            if (e < b || (e > b && !bytes)) return ERH_ERR_INVALID;
            int32_t cnt = 0;
            if (e > b) {                                                 // an empty document has no tokens
                int64_t t0 = b;
                for (int64_t i = b; i <= e; ++i) {
                    if (i == e || (unsigned char)bytes[i] == (unsigned char)sep) {
                        const int32_t id = v->find_or_add(bytes + t0, (size_t)(i - t0), add != 0);
                        if (id == -2) return ERH_ERR_UNSUPPORTED;        // more than 2^31 - 2 distinct tokens
                        if (w < cap && out_ids) out_ids[w] = id;
                        ++w;
                        ++cnt;
                        t0 = i + 1;
                    }
                }
            }
            out_lens[d] = cnt;
        }
    } catch (const std::bad_alloc &) {
        return ERH_ERR_NOMEM;
    }
    *n_out = w;
    return (out_ids && w > cap) ? ERH_ERR_OVERFLOW : ERH_OK;             // (the caller re-runs with a larger buffer; ids are stable)
}

int erh_vocab_token(const erh_vocab *v, int32_t id, const char **bytes, int32_t *len) {
    if (!v || !bytes || !len || id < 0 || (size_t)id >= v->hashes.size()) return ERH_ERR_INVALID;
    *bytes = v->arena.data() + v->tok_off[id];
    *len = (int32_t)(v->tok_off[id + 1] - v->tok_off[id]);
    return ERH_OK;
}

int erh_cutter_create(const char *dict_text, int64_t n_bytes, erh_cutter **out) {
    if (!out || (n_bytes > 0 && !dict_text) || n_bytes < 0) return ERH_ERR_INVALID;
    *out = nullptr;
    erh_cutter *c = new (std::nothrow) erh_cutter();
    if (!c) return ERH_ERR_NOMEM;
    try {
        // jieba.Tokenizer.gen_pfdict: `word, freq = line.split(' ')[:2]`; every prefix of a word enters with frequency 0
        int64_t i = 0;
        std::vector<uint32_t> cp;
        std::vector<int64_t> off;
        while (i < n_bytes) {
            int64_t e = i;
            while (e < n_bytes && dict_text[e] != '\n') ++e;
            int64_t le = e;
            while (le > i && (dict_text[le - 1] == '\r' || dict_text[le - 1] == ' ' || dict_text[le - 1] == '\t')) --le;
            int64_t ls = i;
            while (ls < le && (dict_text[ls] == ' ' || dict_text[ls] == '\t')) ++ls;   // (jieba strips the line)
            if (ls < le) {
                int64_t sp = ls;
                while (sp < le && dict_text[sp] != ' ') ++sp;
                if (sp == le) { delete c; return ERH_ERR_INVALID; }      // "word freq" needs the frequency
                int64_t fe = sp + 1;
                while (fe < le && dict_text[fe] != ' ') ++fe;
                char *endp = nullptr;
                const std::string num(dict_text + sp + 1, (size_t)(fe - sp - 1));
                const long long f = strtoll(num.c_str(), &endp, 10);
                if (num.empty() || *endp != '\0' || f < 0) { delete c; return ERH_ERR_INVALID; }
                const char *wp = dict_text + ls;
                const size_t wn = (size_t)(sp - ls);
                c->set(wp, wn, f, true);                                 // lfreq[word] = freq (a repeated word: the last one)
                c->total += (double)f;
                decode_utf8(wp, (int64_t)wn, cp, off);
                for (size_t ch = 0; ch + 1 < cp.size(); ++ch)            // proper prefixes enter with 0 unless present
                    c->set(wp, (size_t)off[ch + 1], 0, false);
            }
            i = e + 1;
        }
    } catch (const std::bad_alloc &) {
        delete c;
        return ERH_ERR_NOMEM;
    }
    if (!(c->total > 0.0)) { delete c; return ERH_ERR_INVALID; }         // log(total) must exist
    *out = c;
    return ERH_OK;
}

int erh_cutter_destroy(erh_cutter *c) {
    if (!c) return ERH_ERR_INVALID;
    delete c;
    return ERH_OK;
}

int erh_cutter_set_hmm(erh_cutter *c, const char *model_text, int64_t n_bytes) {
    if (!c || n_bytes < 0 || (n_bytes > 0 && !model_text)) return ERH_ERR_INVALID;
    c->has_hmm = false;
    c->hmm_emit.clear();
    for (int a = 0; a < 4; ++a) { c->hmm_start[a] = kHmmMin; for (int b = 0; b < 4; ++b) c->hmm_trans[a][b] = kHmmMin; }
    if (n_bytes == 0) return ERH_OK;                                     // (model removed: HMM=False again)
    auto state_of = [](const std::string &t) -> int {
        if (t.size() != 1) return -1;
        switch (t[0]) { case 'B': return 0; case 'E': return 1; case 'M': return 2; case 'S': return 3; default: return -1; }
    };
    try {
        int64_t i = 0;
        std::vector<uint32_t> cp;
        std::vector<int64_t> off;
        int64_t n_emit = 0;
        while (i < n_bytes) {
            int64_t e = i;
            while (e < n_bytes && model_text[e] != '\n') ++e;
            std::vector<std::string> f;                                  // fields split on single spaces / tabs
            int64_t p = i;
            while (p < e) {
                while (p < e && (model_text[p] == ' ' || model_text[p] == '\t' || model_text[p] == '\r')) ++p;
                int64_t q = p;
                while (q < e && model_text[q] != ' ' && model_text[q] != '\t' && model_text[q] != '\r') ++q;
                if (q > p) f.emplace_back(model_text + p, (size_t)(q - p));
                p = q;
            }
            i = e + 1;
            if (f.empty() || f[0][0] == '#') continue;
            auto num = [&](const std::string &t, double &out) -> bool {
                char *endp = nullptr;
                out = strtod(t.c_str(), &endp);
                return !t.empty() && *endp == '\0';
            };
            double v = 0.0;
            if (f[0] == "start" && f.size() == 3 && state_of(f[1]) >= 0 && num(f[2], v)) {
                c->hmm_start[state_of(f[1])] = v;
            } else if (f[0] == "trans" && f.size() == 4 && state_of(f[1]) >= 0 && state_of(f[2]) >= 0 && num(f[3], v)) {
                c->hmm_trans[state_of(f[1])][state_of(f[2])] = v;
            } else if (f[0] == "emit" && f.size() == 4 && state_of(f[1]) >= 0 && num(f[3], v)) {
                decode_utf8(f[2].data(), (int64_t)f[2].size(), cp, off);
                if (cp.size() != 1) return ERH_ERR_INVALID;
                auto it = c->hmm_emit.find(cp[0]);
                if (it == c->hmm_emit.end()) it = c->hmm_emit.emplace(cp[0], erh_hmm_emit{{kHmmMin, kHmmMin, kHmmMin, kHmmMin}}).first;
                it->second.p[state_of(f[1])] = v;
                ++n_emit;
            } else {
                c->hmm_emit.clear();
                return ERH_ERR_INVALID;
            }
        }
        if (n_emit == 0) return ERH_ERR_INVALID;
    } catch (const std::bad_alloc &) {
        c->hmm_emit.clear();
        return ERH_ERR_NOMEM;
    }
    c->has_hmm = true;
    return ERH_OK;
}

int erh_cutter_has_hmm(const erh_cutter *c) { return (c && c->has_hmm) ? 1 : 0; }

int erh_cutter_cut(const erh_cutter *c, const char *text, int64_t n_bytes, int64_t *out_ends, int64_t cap, int64_t *n_tokens) {
    return erh_cutter_cut_mode(c, text, n_bytes, -1, out_ends, cap, n_tokens);
}

int erh_cutter_cut_mode(const erh_cutter *c, const char *text, int64_t n_bytes, int hmm, int64_t *out_ends, int64_t cap,
                        int64_t *n_tokens) {
    if (!c || !n_tokens || n_bytes < 0 || (n_bytes > 0 && !text) || hmm < -1 || hmm > 1) return ERH_ERR_INVALID;
    if (hmm == 1 && !c->has_hmm) return ERH_ERR_STATE;                   // HMM=True needs the model (erh_cutter_set_hmm)
    const bool use_hmm = hmm < 0 ? c->has_hmm : hmm == 1;
    try {
        std::vector<uint32_t> cp;
        std::vector<int64_t> off, ends;
        cut_text(c, text, n_bytes, cp, off, ends, use_hmm);
        *n_tokens = (int64_t)ends.size();
        if (out_ends) {
            if ((int64_t)ends.size() > cap) return ERH_ERR_OVERFLOW;
            for (size_t t = 0; t < ends.size(); ++t) out_ends[t] = off[(size_t)ends[t]];   // byte offsets of the token ends
        }
    } catch (const std::bad_alloc &) {
        return ERH_ERR_NOMEM;
    }
    return ERH_OK;
}

int erh_text_encode(const erh_cutter *c, erh_vocab *v, const erh_vocab *stop, const char *bytes, const int64_t *text_off,
                    int64_t n_texts, int add, int32_t *out_ids, int64_t cap, int32_t *out_lens, int64_t *n_out) {
    return erh_text_encode_mt(c, v, stop, bytes, text_off, n_texts, add, 1, out_ids, cap, out_lens, n_out);
}

// n_threads > 1 (add != 0 only): the texts are cut into contiguous chunks, one per thread.  Every thread cuts its chunk and
// numbers its tokens in a vocabulary of its own (first appearance inside the chunk); the chunk vocabularies are then
// merged into `v` chunk by chunk, each in its local id order -- which IS first-appearance order over the whole corpus,
// so the ids are those of the one-thread walk -- and every thread rewrites its ids through its local -> global table.
// The dictionary and the stop set are only read.
int erh_text_encode_mt(const erh_cutter *c, erh_vocab *v, const erh_vocab *stop, const char *bytes, const int64_t *text_off,
                       int64_t n_texts, int add, int n_threads, int32_t *out_ids, int64_t cap, int32_t *out_lens,
                       int64_t *n_out) {
    if (!c || !v || !text_off || n_texts < 0 || !n_out || (n_texts > 0 && !out_lens) || n_threads < 1) return ERH_ERR_INVALID;
    for (int64_t d = 0; d < n_texts; ++d)
        if (text_off[d + 1] < text_off[d] || (text_off[d + 1] > text_off[d] && !bytes)) return ERH_ERR_INVALID;
    // one chunk: tokens of texts [d0, d1) through `voc` (add) -> ids appended to `ids`, token counts to out_lens
    auto run = [&](int64_t d0, int64_t d1, erh_vocab *voc, bool add_, std::vector<int32_t> &ids) -> int {
        std::vector<uint32_t> cp;
        std::vector<int64_t> off, ends;
        for (int64_t d = d0; d < d1; ++d) {
            const int64_t b = text_off[d], e = text_off[d + 1];
            cut_text(c, bytes + b, e - b, cp, off, ends, c->has_hmm);
            int32_t cnt = 0;
            int64_t t0 = 0;
            for (int64_t end_ch : ends) {
                const int64_t t1 = off[(size_t)end_ch];
                const char *tp = bytes + b + t0;
                const size_t tn = (size_t)(t1 - t0);
                t0 = t1;
                if (tn == 1 && tp[0] == ' ') continue;                   // `word != ' '` (retrievers.py:75)
                if (stop && const_cast<erh_vocab *>(stop)->find_or_add(tp, tn, false) >= 0) continue;   // `word not in stopwords`
                const int32_t id = voc->find_or_add(tp, tn, add_);
                if (id == -2) return ERH_ERR_UNSUPPORTED;
                if (id < 0) continue;                                    // add == 0: out-of-vocabulary query token
                ids.push_back(id);
                ++cnt;
            }
            out_lens[d] = cnt;
        }
        return ERH_OK;
    };
    int64_t w = 0;
    try {
        int T = n_threads;
        if (!add || n_texts < 2 * (int64_t)T) T = 1;                     // (queries: unknown tokens are dropped -- one thread)
        if (T == 1) {
            std::vector<int32_t> ids;
            const int rc = run(0, n_texts, v, add != 0, ids);
            if (rc != ERH_OK) return rc;
            w = (int64_t)ids.size();
            if (out_ids) memcpy(out_ids, ids.data(), (size_t)(w < cap ? w : cap) * 4);
        } else {
            // chunks of about equal bytes
            std::vector<int64_t> cut((size_t)T + 1, n_texts);
            cut[0] = 0;
            const int64_t total = text_off[n_texts] - text_off[0];
            for (int t = 1, d = 0; t < T; ++t) {
                const int64_t want = text_off[0] + total * t / T;
                while (d < n_texts && text_off[d] < want) ++d;
                cut[(size_t)t] = d;
            }
            std::vector<erh_vocab> local((size_t)T);
            std::vector<std::vector<int32_t>> ids((size_t)T);
            std::vector<int> rcs((size_t)T, ERH_OK);
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t]() {
                    try { rcs[(size_t)t] = run(cut[(size_t)t], cut[(size_t)t + 1], &local[(size_t)t], true, ids[(size_t)t]); }
                    catch (const std::bad_alloc &) { rcs[(size_t)t] = ERH_ERR_NOMEM; }
                });
            for (auto &x : th) x.join();
            for (int t = 0; t < T; ++t) if (rcs[(size_t)t] != ERH_OK) return rcs[(size_t)t];
            // merge the chunk vocabularies in chunk order, each in its local id order
            std::vector<std::vector<int32_t>> map((size_t)T);
            for (int t = 0; t < T; ++t) {
                const erh_vocab &lv = local[(size_t)t];
                map[(size_t)t].resize(lv.hashes.size());
                for (size_t id = 0; id < lv.hashes.size(); ++id) {
                    const int64_t b = lv.tok_off[id], e = lv.tok_off[id + 1];
                    const int32_t g = v->find_or_add(lv.hashes[id], lv.arena.data() + b, (size_t)(e - b), true);
                    if (g == -2) return ERH_ERR_UNSUPPORTED;
                    map[(size_t)t][id] = g;
                }
            }
            std::vector<int64_t> base((size_t)T + 1, 0);
            for (int t = 0; t < T; ++t) base[(size_t)t + 1] = base[(size_t)t] + (int64_t)ids[(size_t)t].size();
            w = base[(size_t)T];
            if (out_ids) {
                th.clear();
                for (int t = 0; t < T; ++t)
                    th.emplace_back([&, t]() {
                        const std::vector<int32_t> &src = ids[(size_t)t];
                        const std::vector<int32_t> &m = map[(size_t)t];
                        const int64_t o = base[(size_t)t];
                        for (size_t i = 0; i < src.size() && o + (int64_t)i < cap; ++i) out_ids[o + (int64_t)i] = m[(size_t)src[i]];
                    });
                for (auto &x : th) x.join();
            }
        }
    } catch (const std::bad_alloc &) {
        return ERH_ERR_NOMEM;
    } catch (const std::system_error &) {
        return ERH_ERR_NOMEM;                                            // (no thread to be had)
    }
    *n_out = w;
    return (out_ids && w > cap) ? ERH_ERR_OVERFLOW : ERH_OK;
}

}  // extern "C"
