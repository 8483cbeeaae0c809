// Sanitizer driver for the host-only text side (easyrag_amd/csrc/text.hip compiled as plain C++ with
// -fsanitize=address,undefined): random dictionaries, HMM models and byte strings through every entry point of the text
// ABI, one and several threads.  Built and run by tests/test_text_sanitizers.py; prints "ok <n>" on success.
#include "../../include/easyrag_hip.h"

#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

static std::string utf8(uint32_t c) {
    std::string s;
    if (c < 0x80) s += (char)c;
    else if (c < 0x800) { s += (char)(0xC0 | (c >> 6)); s += (char)(0x80 | (c & 0x3F)); }
    else { s += (char)(0xE0 | (c >> 12)); s += (char)(0x80 | ((c >> 6) & 0x3F)); s += (char)(0x80 | (c & 0x3F)); }
    return s;
}

#define CHECK(cond) do { if (!(cond)) { std::fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #cond); return 1; } } while (0)

int main() {
    std::mt19937 rng(12345);
    auto rnd = [&](int n) { return (int)(rng() % (uint32_t)n); };
    long checked = 0;
    for (int round = 0; round < 30; ++round) {
        std::vector<uint32_t> alpha;
        for (int i = 0; i < 30; ++i) alpha.push_back(0x4E00 + rnd(200));
        std::string dict;
        for (int w = 0; w < 40; ++w) {
            const int len = 1 + rnd(4);
            for (int i = 0; i < len; ++i) dict += utf8(alpha[rnd((int)alpha.size())]);
            dict += " " + std::to_string(1 + rnd(900)) + (rnd(3) ? "" : " n") + (rnd(5) ? "\n" : "\r\n");
        }
        dict += "C++ 7\nT恤 3\n";
        erh_cutter *c = nullptr;
        CHECK(erh_cutter_create(dict.data(), (int64_t)dict.size(), &c) == ERH_OK);
        if (round % 2) {
            std::string m = "start B -0.3\nstart S -1.4\ntrans B E -0.5\ntrans B M -0.9\ntrans E B -0.6\ntrans E S -0.8\n"
                            "trans M E -0.3\ntrans M M -1.3\ntrans S B -0.7\ntrans S S -0.7\n";
            const char st[4] = {'B', 'E', 'M', 'S'};
            for (uint32_t ch : alpha)
                for (int s = 0; s < 4; ++s)
                    if (rnd(5)) m += std::string("emit ") + st[s] + " " + utf8(ch) + " -" + std::to_string(2 + rnd(12)) + ".25\n";
            CHECK(erh_cutter_set_hmm(c, m.data(), (int64_t)m.size()) == ERH_OK);
            CHECK(erh_cutter_has_hmm(c) == 1);
        }
        // texts: dictionary characters, ASCII, white space, malformed UTF-8
        std::vector<std::string> texts;
        for (int t = 0; t < 60; ++t) {
            std::string s;
            const int n = rnd(40);
            for (int i = 0; i < n; ++i) {
                const int k = rnd(10);
                if (k < 6) s += utf8(alpha[rnd((int)alpha.size())]);
                else if (k == 6) s += "a1.5%+C#"[rnd(8)];
                else if (k == 7) s += " \t\r\n"[rnd(4)];
                else if (k == 8) s += (char)(0x80 + rnd(128));
                else s += utf8(0x3000 + rnd(64));
            }
            texts.push_back(s);
        }
        std::string blob;
        std::vector<int64_t> off{0};
        for (auto &t : texts) { blob += t; off.push_back((int64_t)blob.size()); }
        for (auto &t : texts) {
            std::vector<int64_t> ends(t.size() + 1);
            int64_t n = 0;
            for (int mode = -1; mode <= (round % 2 ? 1 : 0); ++mode) {
                CHECK(erh_cutter_cut_mode(c, t.data(), (int64_t)t.size(), mode, ends.data(), (int64_t)ends.size(), &n) == ERH_OK);
                int64_t prev = 0;
                for (int64_t i = 0; i < n; ++i) { CHECK(ends[i] > prev); prev = ends[i]; }
                CHECK(prev == (int64_t)t.size());
                ++checked;
            }
            CHECK(erh_cutter_cut(c, t.data(), (int64_t)t.size(), nullptr, 0, &n) == ERH_OK);   // count only
        }
        if (!(round % 2)) {
            int64_t n = 0;
            CHECK(erh_cutter_cut_mode(c, "x", 1, 1, nullptr, 0, &n) == ERH_ERR_STATE);
        }
        // fused encode: thread counts agree
        std::vector<std::vector<int32_t>> ids_by_t;
        for (int threads : {1, 2, 5}) {
            erh_vocab *v = nullptr, *stop = nullptr;
            CHECK(erh_vocab_create(&v) == ERH_OK && erh_vocab_create(&stop) == ERH_OK);
            const std::string sw = utf8(alpha[0]);
            const int64_t so[2] = {0, (int64_t)sw.size()};
            int32_t sl = 0;
            int64_t sn = 0;
            int32_t sid[4];
            CHECK(erh_vocab_encode(stop, sw.data(), so, 1, 0x1f, 1, sid, 4, &sl, &sn) == ERH_OK);
            std::vector<int32_t> ids(blob.size() + 1), lens(texts.size());
            int64_t need = 0;
            CHECK(erh_text_encode_mt(c, v, stop, blob.data(), off.data(), (int64_t)texts.size(), 1, threads, ids.data(),
                                     (int64_t)ids.size(), lens.data(), &need) == ERH_OK);
            ids.resize((size_t)need);
            ids_by_t.push_back(ids);
            for (int32_t id : ids) {
                const char *tp = nullptr;
                int32_t tl = 0;
                CHECK(erh_vocab_token(v, id, &tp, &tl) == ERH_OK && tl > 0);
            }
            // query side
            int64_t qn = 0;
            int32_t ql = 0;
            std::vector<int32_t> qids(texts[0].size() + 1);
            const int64_t qo[2] = {0, (int64_t)texts[0].size()};
            CHECK(erh_text_encode(c, v, stop, texts[0].data(), qo, 1, 0, qids.data(), (int64_t)qids.size(), &ql, &qn) == ERH_OK);
            erh_vocab_destroy(v);
            erh_vocab_destroy(stop);
        }
        CHECK(ids_by_t[0] == ids_by_t[1] && ids_by_t[0] == ids_by_t[2]);
        // malformed inputs are refused, not crashed on
        erh_cutter *bad = nullptr;
        CHECK(erh_cutter_create("word\n", 5, &bad) == ERH_ERR_INVALID);
        CHECK(erh_cutter_set_hmm(c, "emit Q x -1\n", 12) == ERH_ERR_INVALID);
        CHECK(erh_cutter_set_hmm(c, "", 0) == ERH_OK && erh_cutter_has_hmm(c) == 0);
        erh_cutter_destroy(c);
    }
    std::printf("ok %ld\n", checked);
    return 0;
}
