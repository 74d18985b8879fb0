// Host-side BERT WordPiece tokenizer for ASCII sentences (C ABI: include/tsim.h "tokenizer").
//
// The reference tokenises on the host with the HuggingFace tokenizer it is configured with
// (/root/reference/src/models/sentence_encoder.py:144-153: tokenizer(text=..., padding=True, truncation=True,
// max_length=...)), and once the encoder runs on an MI355X that call is the end-to-end bottleneck of encode_text
// (16 384 sentences: 70 ms in the tokenizer, 10 ms on the GPU).  This file restates, for sentences
// that are pure ASCII, the pipeline of the `tokenizers` library's BERT configuration (third-party dependency of the
// reference, not vendored in it; the published algorithm: BertNormalizer -> BertPreTokenizer -> WordPiece -> template
// post-processor -> truncation):
//   normalizer    clean_text: drop NUL and control characters (Cc: 0x01-0x1F except \t \n \r, and 0x7F), map \t \n \r to ' ';
//                 lowercase (optional).  Accent stripping and CJK spacing do nothing to ASCII.
//   pre-tokenizer split on whitespace; every ASCII punctuation character (33-47, 58-64, 91-96, 123-126) is a word of its own.
//   WordPiece     greedy longest-match-first with the continuing-subword prefix; a word longer than max_input_chars_per_word
//                 or with an unmatched remainder is ONE unk token.
//   post          prefix ids (CLS) + at most max_len - specials word pieces (truncation on the right) + suffix ids (SEP).
// A sentence that is not pure ASCII, or that contains the text of an added / special token (matched by the library on the
// raw text before normalisation), is NOT handled: handled[i] = 0 and the caller runs it through the library itself.
// The library stays the oracle: tests/test_wordpiece_cpu.py compares ids on generated and hand-written sentences.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "tsim.h"

namespace {

struct Vocab {
    // open addressing, keys are slices of one blob
    std::vector<char> blob;
    std::vector<uint32_t> off, len;   // per entry
    std::vector<int32_t> id;
    std::vector<int32_t> slot;        // table -> entry or -1
    uint32_t mask = 0;
    uint32_t max_len = 0;

    static uint64_t hash(const char *p, size_t n) {
        uint64_t h = 1469598103934665603ull;   // FNV-1a, finished with a mix
        for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)p[i]; h *= 1099511628211ull; }
        h ^= h >> 29; h *= 0xbf58476d1ce4e5b9ull; h ^= h >> 32;
        return h;
    }
    void build(const char *text, const int64_t *offsets, int32_t n) {
        blob.assign(text, text + offsets[n]);
        off.resize(n); len.resize(n); id.resize(n);
        size_t cap = 16;
        while (cap < (size_t)n * 2 + 2) cap <<= 1;
        slot.assign(cap, -1);
        mask = (uint32_t)(cap - 1);
        for (int32_t i = 0; i < n; ++i) {
            off[i] = (uint32_t)offsets[i];
            len[i] = (uint32_t)(offsets[i + 1] - offsets[i]);
            id[i] = i;
            max_len = std::max(max_len, len[i]);
            uint32_t s = (uint32_t)hash(blob.data() + off[i], len[i]) & mask;
            bool dup = false;
            while (slot[s] >= 0) {   // a later duplicate of a key keeps the first entry's slot but takes its id (dict semantics)
                const int32_t e = slot[s];
                if (len[e] == len[i] && memcmp(blob.data() + off[e], blob.data() + off[i], len[i]) == 0) { id[e] = i; dup = true; break; }
                s = (s + 1) & mask;
            }
            if (!dup) slot[s] = i;
        }
    }
    int32_t find(const char *p, size_t n) const {
        if (n > max_len) return -1;
        uint32_t s = (uint32_t)hash(p, n) & mask;
        while (slot[s] >= 0) {
            const int32_t e = slot[s];
            if (len[e] == n && memcmp(blob.data() + off[e], p, n) == 0) return id[e];
            s = (s + 1) & mask;
        }
        return -1;
    }
};

struct WordPiece {
    Vocab whole, cont;   // entries without / with the continuing-subword prefix (prefix stripped in `cont`)
    int32_t unk_id = 0;
    std::vector<int32_t> prefix_ids, suffix_ids;
    bool lowercase = true;
    int max_chars = 100;
    std::vector<std::string> added;   // added-token contents: a sentence containing one is not handled
};

inline bool is_punct(unsigned char c) { return (c >= 33 && c <= 47) || (c >= 58 && c <= 64) || (c >= 91 && c <= 96) || (c >= 123 && c <= 126); }

// word pieces of one normalised word [w, w + n) appended to out (at most `room` are kept; returns how many it produced, which
// may exceed room: the caller truncates)
inline void wordpiece_word(const WordPiece &wp, const char *w, int n, std::vector<int32_t> &out) {
    if (n > wp.max_chars) { out.push_back(wp.unk_id); return; }
    const size_t mark = out.size();
    int start = 0;
    while (start < n) {
        const Vocab &v = start == 0 ? wp.whole : wp.cont;
        int end = std::min<int>(n, start + (int)v.max_len);
        int32_t hit = -1;
        for (; end > start; --end) {
            hit = v.find(w + start, (size_t)(end - start));
            if (hit >= 0) break;
        }
        if (hit < 0) { out.resize(mark); out.push_back(wp.unk_id); return; }
        out.push_back(hit);
        start = end;
    }
}

// one sentence -> ids (specials included, truncated to max_len); false: not handled
bool encode_one(const WordPiece &wp, const char *s, int64_t n, int max_len, std::vector<int32_t> &pieces, std::vector<char> &norm,
                int32_t *out, int32_t *out_len) {
    for (int64_t i = 0; i < n; ++i)
        if ((unsigned char)s[i] >= 0x80) return false;
    for (const std::string &a : wp.added)
        if (!a.empty() && (int64_t)a.size() <= n && std::search(s, s + n, a.begin(), a.end()) != s + n) return false;
    const int nspec = (int)(wp.prefix_ids.size() + wp.suffix_ids.size());
    const int room = max_len > nspec ? max_len - nspec : 0;
    // normalise
    norm.clear();
    for (int64_t i = 0; i < n; ++i) {
        unsigned char c = (unsigned char)s[i];
        if (c == '\t' || c == '\n' || c == '\r') c = ' ';
        else if (c < 0x20 || c == 0x7f) continue;
        if (wp.lowercase && c >= 'A' && c <= 'Z') c = (unsigned char)(c + 32);
        norm.push_back((char)c);
    }
    pieces.clear();
    const int m = (int)norm.size();
    int i = 0;
    while (i < m && (int)pieces.size() < room) {   // pieces beyond `room` are cut anyway: stop early
        const unsigned char c = (unsigned char)norm[i];
        if (c == ' ') { ++i; continue; }
        if (is_punct(c)) { wordpiece_word(wp, norm.data() + i, 1, pieces); ++i; continue; }
        int j = i + 1;
        while (j < m && norm[j] != ' ' && !is_punct((unsigned char)norm[j])) ++j;
        wordpiece_word(wp, norm.data() + i, j - i, pieces);
        i = j;
    }
    const int keep = std::min<int>((int)pieces.size(), room);
    int o = 0;
    for (int32_t t : wp.prefix_ids) out[o++] = t;
    for (int t = 0; t < keep; ++t) out[o++] = pieces[t];
    for (int32_t t : wp.suffix_ids) out[o++] = t;
    *out_len = o;
    return true;
}

}  // namespace

extern "C" int tsim_wordpiece_create(const char *vocab_text, const int64_t *vocab_offsets, int32_t vocab_size,
                                     const char *continuing_prefix, int32_t unk_id, const int32_t *prefix_ids, int32_t n_prefix,
                                     const int32_t *suffix_ids, int32_t n_suffix, int32_t lowercase,
                                     int32_t max_input_chars_per_word, const char *added_text, const int64_t *added_offsets,
                                     int32_t n_added, void **handle) {
    if (!vocab_text || !vocab_offsets || vocab_size <= 0 || !continuing_prefix || !handle || n_prefix < 0 || n_suffix < 0 ||
        (n_prefix && !prefix_ids) || (n_suffix && !suffix_ids) || (n_added && (!added_text || !added_offsets)))
        return TSIM_EINVAL;
    WordPiece *wp = new WordPiece();
    const size_t pl = strlen(continuing_prefix);
    // split the vocabulary: ids are positions in the caller's list
    std::vector<char> wb, cb;
    std::vector<int64_t> wo{0}, co{0};
    std::vector<int32_t> wid, cid;
    for (int32_t i = 0; i < vocab_size; ++i) {
        const char *p = vocab_text + vocab_offsets[i];
        const size_t n = (size_t)(vocab_offsets[i + 1] - vocab_offsets[i]);
        // a key that starts with the prefix can still be matched as a whole word (e.g. the word "##" itself is punctuation and
        // never reaches the matcher as one word, but keep the library's behaviour: the whole-word table holds every key)
        wb.insert(wb.end(), p, p + n); wo.push_back((int64_t)wb.size()); wid.push_back(i);
        if (pl && n > pl && memcmp(p, continuing_prefix, pl) == 0) {
            cb.insert(cb.end(), p + pl, p + n); co.push_back((int64_t)cb.size()); cid.push_back(i);
        } else if (!pl) {
            cb.insert(cb.end(), p, p + n); co.push_back((int64_t)cb.size()); cid.push_back(i);
        }
    }
    wp->whole.build(wb.data(), wo.data(), (int32_t)wid.size());
    for (size_t e = 0; e < wp->whole.id.size(); ++e) wp->whole.id[e] = wid[wp->whole.id[e]];
    if (!cid.empty()) {
        wp->cont.build(cb.data(), co.data(), (int32_t)cid.size());
        for (size_t e = 0; e < wp->cont.id.size(); ++e) wp->cont.id[e] = cid[wp->cont.id[e]];
    } else {
        wp->cont.slot.assign(16, -1);
        wp->cont.mask = 15;
    }
    wp->unk_id = unk_id;
    wp->prefix_ids.assign(prefix_ids, prefix_ids + n_prefix);
    wp->suffix_ids.assign(suffix_ids, suffix_ids + n_suffix);
    wp->lowercase = lowercase != 0;
    wp->max_chars = max_input_chars_per_word;
    for (int32_t i = 0; i < n_added; ++i)
        wp->added.emplace_back(added_text + added_offsets[i], added_text + added_offsets[i + 1]);
    *handle = wp;
    return TSIM_OK;
}

extern "C" void tsim_wordpiece_destroy(void *handle) { delete static_cast<WordPiece *>(handle); }

extern "C" int tsim_wordpiece_encode(void *handle, const char *text, const int64_t *text_offsets, int64_t n, int32_t max_len,
                                     int32_t n_threads, int32_t *out_ids, int64_t out_capacity, int32_t *out_lens,
                                     uint8_t *handled) {
    if (!handle || !text_offsets || n < 0 || max_len <= 0 || !out_ids || !out_lens || !handled || (n && !text)) return TSIM_EINVAL;
    const WordPiece &wp = *static_cast<const WordPiece *>(handle);
    if (n == 0) return TSIM_OK;
    // a word piece consumes at least one byte of its sentence: row i needs at most min(max_len, bytes + specials) ids
    const int64_t nspec = (int64_t)(wp.prefix_ids.size() + wp.suffix_ids.size());
    std::vector<int64_t> start((size_t)n + 1);
    start[0] = 0;
    for (int64_t i = 0; i < n; ++i)
        start[i + 1] = start[i] + std::max<int64_t>(nspec, std::min<int64_t>(max_len, text_offsets[i + 1] - text_offsets[i] + nspec));
    if (out_capacity < start[n]) return TSIM_ENOMEM;
    int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    nt = std::max(1, std::min<int>(nt, (int)std::min<int64_t>((n + 63) / 64, 64)));
    // sentences are dealt out in blocks of 64 (sorted input has its long sentences together: contiguous ranges would be uneven)
    auto work = [&](int t) {
        std::vector<int32_t> pieces;
        std::vector<char> norm;
        for (int64_t b = (int64_t)t * 64; b < n; b += (int64_t)nt * 64)
            for (int64_t i = b; i < std::min<int64_t>(b + 64, n); ++i) {
                const bool ok = encode_one(wp, text + text_offsets[i], text_offsets[i + 1] - text_offsets[i], max_len, pieces, norm,
                                           out_ids + start[i], out_lens + i);
                handled[i] = ok ? 1 : 0;
                if (!ok) out_lens[i] = 0;
            }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    // compact in place (destination never passes the source)
    int64_t o = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (o != start[i] && out_lens[i]) memmove(out_ids + o, out_ids + start[i], (size_t)out_lens[i] * sizeof(int32_t));
        o += out_lens[i];
    }
    return TSIM_OK;
}
