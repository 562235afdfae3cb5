// kai_ingest.cpp — reference-schema snapshot.json / snapshot.zip → kai_snapshot_soa + kai_config (include/kai_ingest.h).
//
// Host-only data-format conversion (SURVEY.md §8f n1 + n4).  Every rule below cites the reference code it restates; all paths are
// under /root/reference/pkg/scheduler unless they start with cmd/ or pkg/.  Nothing here decides a placement: the output is the
// input of kai_session_open.
#include "../../include/kai_ingest.h"

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <deque>
#include <map>
#include <set>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

thread_local std::string g_err;
typedef __int128 i128;

// ===================================================================================================== JSON (RFC 8259) DOM
struct JV {
    enum T : uint8_t { Null, Bool, Num, Str, Arr, Obj } t = Null;
    bool b = false;
    std::string s;  // Str: decoded text; Num: the literal
    std::vector<JV> a;
    std::vector<std::pair<std::string, JV>> o;
    const JV& operator[](const char* k) const;  // object member (the last duplicate wins, like encoding/json), or null
    bool is_obj() const { return t == Obj; }
    bool is_arr() const { return t == Arr; }
    bool is_null() const { return t == Null; }
    const std::string& str() const { static const std::string e; return t == Str ? s : e; }
    double num() const { return t == Num ? strtod(s.c_str(), nullptr) : 0.0; }
    int64_t inum() const { return t == Num ? (int64_t)strtoll(s.c_str(), nullptr, 10) : 0; }
    bool truthy() const { return t == Bool && b; }
};
const JV kNull;
const JV& JV::operator[](const char* k) const {
    if (t != Obj) return kNull;
    for (size_t i = o.size(); i-- > 0;) if (o[i].first == k) return o[i].second;
    return kNull;
}

struct JParser {
    const char* p; const char* e; std::string err; int depth = 0;
    void ws() { while (p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++; }
    bool fail(const char* m) { if (err.empty()) err = m; return false; }
    static void utf8(std::string& out, uint32_t c) {
        if (c < 0x80) out += (char)c;
        else if (c < 0x800) { out += (char)(0xC0 | (c >> 6)); out += (char)(0x80 | (c & 0x3F)); }
        else if (c < 0x10000) { out += (char)(0xE0 | (c >> 12)); out += (char)(0x80 | ((c >> 6) & 0x3F)); out += (char)(0x80 | (c & 0x3F)); }
        else { out += (char)(0xF0 | (c >> 18)); out += (char)(0x80 | ((c >> 12) & 0x3F)); out += (char)(0x80 | ((c >> 6) & 0x3F)); out += (char)(0x80 | (c & 0x3F)); }
    }
    bool hex4(uint32_t& v) {
        if (e - p < 4) return fail("truncated \\u escape");
        v = 0;
        for (int i = 0; i < 4; i++) { char c = *p++; v <<= 4; if (c >= '0' && c <= '9') v |= c - '0'; else if (c >= 'a' && c <= 'f') v |= c - 'a' + 10; else if (c >= 'A' && c <= 'F') v |= c - 'A' + 10; else return fail("bad \\u escape"); }
        return true;
    }
    bool string(std::string& out) {
        if (p >= e || *p != '"') return fail("expected string");
        p++;
        for (;;) {
            const char* q = p;
            while (q < e && *q != '"' && *q != '\\') q++;
            out.append(p, q - p); p = q;
            if (p >= e) return fail("unterminated string");
            if (*p == '"') { p++; return true; }
            p++;
            if (p >= e) return fail("unterminated escape");
            char c = *p++;
            switch (c) {
                case '"': out += '"'; break; case '\\': out += '\\'; break; case '/': out += '/'; break;
                case 'b': out += '\b'; break; case 'f': out += '\f'; break; case 'n': out += '\n'; break; case 'r': out += '\r'; break; case 't': out += '\t'; break;
                case 'u': {
                    uint32_t v; if (!hex4(v)) return false;
                    if (v >= 0xD800 && v < 0xDC00 && e - p >= 6 && p[0] == '\\' && p[1] == 'u') { p += 2; uint32_t lo; if (!hex4(lo)) return false; if (lo >= 0xDC00 && lo < 0xE000) v = 0x10000 + ((v - 0xD800) << 10) + (lo - 0xDC00); else v = 0xFFFD; }
                    utf8(out, v); break;
                }
                default: return fail("bad escape");
            }
        }
    }
    bool value(JV& v) {
        ws();
        if (p >= e) return fail("unexpected end of input");
        if (++depth > 256) return fail("nesting too deep");
        bool ok = true;
        char c = *p;
        if (c == '{') {
            v.t = JV::Obj; p++; ws();
            if (p < e && *p == '}') p++;
            else for (;;) {
                ws(); std::string k; if (!string(k)) { ok = false; break; }
                ws(); if (p >= e || *p != ':') { ok = fail("expected ':'"); break; } p++;
                v.o.emplace_back(std::move(k), JV()); if (!value(v.o.back().second)) { ok = false; break; }
                ws(); if (p < e && *p == ',') { p++; continue; } if (p < e && *p == '}') { p++; break; } ok = fail("expected ',' or '}'"); break;
            }
        } else if (c == '[') {
            v.t = JV::Arr; p++; ws();
            if (p < e && *p == ']') p++;
            else for (;;) {
                v.a.emplace_back(); if (!value(v.a.back())) { ok = false; break; }
                ws(); if (p < e && *p == ',') { p++; continue; } if (p < e && *p == ']') { p++; break; } ok = fail("expected ',' or ']'"); break;
            }
        } else if (c == '"') { v.t = JV::Str; ok = string(v.s); }
        else if (c == 't' && e - p >= 4 && !memcmp(p, "true", 4)) { v.t = JV::Bool; v.b = true; p += 4; }
        else if (c == 'f' && e - p >= 5 && !memcmp(p, "false", 5)) { v.t = JV::Bool; v.b = false; p += 5; }
        else if (c == 'n' && e - p >= 4 && !memcmp(p, "null", 4)) { v.t = JV::Null; p += 4; }
        else if (c == '-' || (c >= '0' && c <= '9')) {
            const char* q = p; if (*q == '-') q++;
            while (q < e && ((*q >= '0' && *q <= '9') || *q == '.' || *q == 'e' || *q == 'E' || *q == '+' || *q == '-')) q++;
            v.t = JV::Num; v.s.assign(p, q - p); p = q;
        } else ok = fail("unexpected character");
        depth--;
        return ok;
    }
};

// Structural skip of one value (strings with escapes, nesting by depth): finds the element boundaries of rawObjects.pods so that the pods —
// nearly all of a snapshot's bytes — are parsed on every host core.
bool skip_value(const char*& p, const char* e) {
    auto ws = [&] { while (p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++; };
    ws(); if (p >= e) return false;
    if (*p == '"') { p++; while (p < e && *p != '"') { if (*p == '\\') p++; p++; } if (p >= e) return false; p++; return true; }
    if (*p == '{' || *p == '[') {
        int depth = 0;
        while (p < e) {
            char c = *p;
            if (c == '"') { p++; while (p < e && *p != '"') { if (*p == '\\') p++; p++; } if (p >= e) return false; p++; continue; }
            if (c == '{' || c == '[') depth++;
            else if (c == '}' || c == ']') { depth--; if (depth == 0) { p++; return true; } }
            p++;
        }
        return false;
    }
    while (p < e && *p != ',' && *p != '}' && *p != ']' && *p != ' ' && *p != '\n' && *p != '\r' && *p != '\t') p++;
    return true;
}
template <class F> void parallel_for(size_t n, F f);
// the document: {"config":…, "schedulerParams":…, "rawObjects":{"pods":[…], …}, …} — everything through JParser, the pods element-wise in parallel
// rawObjects.pods is by far the largest part of a snapshot (10^6 objects at BASELINE config 5): its elements are only located here (spans) and converted one at a time,
// in parallel, each through a DOM of its own that lives for that one conversion (kai_ingest::build) — the document tree holds everything else
struct PodSpans { std::vector<std::pair<const char*, const char*>> spans; const char* base = nullptr; };
bool parse_document(const char* json, size_t len, JV& root, std::string& err, size_t& err_at, PodSpans* lazy = nullptr) {
    JParser ps{json, json + len, {}, 0};
    auto fail = [&](const char* m) { err = ps.err.empty() ? m : ps.err; err_at = (size_t)(ps.p - json); return false; };
    ps.ws(); if (ps.p >= ps.e || *ps.p != '{') { if (!ps.value(root)) return fail("bad document"); ps.ws(); if (ps.p != ps.e) return fail("trailing data"); return true; }
    root.t = JV::Obj; ps.p++; ps.ws();
    std::vector<std::pair<const char*, const char*>> spans; JV* pods_arr = nullptr;
    if (ps.p < ps.e && *ps.p == '}') ps.p++;
    else for (;;) {
        ps.ws(); std::string k; if (!ps.string(k)) return fail("expected string"); ps.ws(); if (ps.p >= ps.e || *ps.p != ':') return fail("expected ':'"); ps.p++; ps.ws();
        root.o.emplace_back(k, JV()); JV& v = root.o.back().second;
        if (k == "rawObjects" && ps.p < ps.e && *ps.p == '{') {
            v.t = JV::Obj; ps.p++; ps.ws();
            if (ps.p < ps.e && *ps.p == '}') ps.p++;
            else for (;;) {
                ps.ws(); std::string k2; if (!ps.string(k2)) return fail("expected string"); ps.ws(); if (ps.p >= ps.e || *ps.p != ':') return fail("expected ':'"); ps.p++; ps.ws();
                v.o.emplace_back(k2, JV()); JV& v2 = v.o.back().second;
                if (k2 == "pods" && ps.p < ps.e && *ps.p == '[' && !pods_arr) {
                    v2.t = JV::Arr; ps.p++; ps.ws(); spans.clear();
                    if (ps.p < ps.e && *ps.p == ']') ps.p++;
                    else for (;;) {
                        ps.ws(); const char* b = ps.p; if (!skip_value(ps.p, ps.e)) return fail("unterminated pods array"); spans.emplace_back(b, ps.p);
                        ps.ws(); if (ps.p < ps.e && *ps.p == ',') { ps.p++; continue; } if (ps.p < ps.e && *ps.p == ']') { ps.p++; break; } return fail("expected ',' or ']'");
                    }
                } else if (!ps.value(v2)) return fail("bad value");
                ps.ws(); if (ps.p < ps.e && *ps.p == ',') { ps.p++; continue; } if (ps.p < ps.e && *ps.p == '}') { ps.p++; break; } return fail("expected ',' or '}'");
            }
        } else if (!ps.value(v)) return fail("bad value");
        ps.ws(); if (ps.p < ps.e && *ps.p == ',') { ps.p++; continue; } if (ps.p < ps.e && *ps.p == '}') { ps.p++; break; } return fail("expected ',' or '}'");
    }
    ps.ws(); if (ps.p != ps.e) return fail("trailing data");
    for (auto& kv : root.o) if (kv.first == "rawObjects") for (auto& kv2 : kv.second.o) if (kv2.first == "pods" && kv2.second.t == JV::Arr && kv2.second.a.empty() && !pods_arr) pods_arr = &kv2.second;  // addresses are final now
    if (pods_arr && !spans.empty() && lazy) { lazy->spans.swap(spans); lazy->base = json; return true; }
    if (pods_arr && !spans.empty()) {
        pods_arr->a.resize(spans.size()); std::vector<std::string> errs(spans.size()); std::vector<size_t> at(spans.size());
        parallel_for(spans.size(), [&](size_t i) {
            JParser q{spans[i].first, spans[i].second, {}, 0};
            if (!q.value(pods_arr->a[i])) { errs[i] = q.err.empty() ? "bad value" : q.err; at[i] = (size_t)(q.p - json); return; }
            q.ws(); if (q.p != q.e) { errs[i] = "unexpected character"; at[i] = (size_t)(q.p - json); }
        });
        for (size_t i = 0; i < errs.size(); i++) if (!errs[i].empty()) { err = errs[i]; err_at = at[i]; return false; }
    }
    return true;
}

// ===================================================================================================== resource.Quantity
// k8s.io/apimachinery v0.34.3 pkg/api/resource: <sign><digits>[.<digits>]<suffix>, suffix = Ki Mi Gi Ti Pi Ei | n u m "" k M G T P E |
// e<exp> E<exp>.  Held exactly as num / 10^dexp; Value() and MilliValue() round UP (quantity.go ScaledValue → infScale / int64Amount).
struct Qty { i128 num = 0; int dexp = 0; bool ok = true; };  // value = num / 10^dexp
i128 pow10i(int k) { i128 r = 1; while (k-- > 0) r *= 10; return r; }
Qty parse_qty(const std::string& s) {
    Qty q; const char* p = s.c_str(); const char* e = p + s.size();
    while (p < e && *p == ' ') p++;
    bool neg = false; if (p < e && (*p == '+' || *p == '-')) { neg = *p == '-'; p++; }
    int digits = 0, frac = 0; bool dot = false;
    for (; p < e; p++) {
        if (*p >= '0' && *p <= '9') { if (digits < 36) { q.num = q.num * 10 + (*p - '0'); if (dot) frac++; } else if (!dot) { q.ok = false; } digits++; }
        else if (*p == '.' && !dot) dot = true;
        else break;
    }
    if (digits == 0) { q.ok = false; return q; }
    q.dexp = frac;
    std::string suf(p, e - p);
    int e10 = 0, e2 = 0;
    if (suf.empty()) {}
    else if (suf == "Ki") e2 = 10; else if (suf == "Mi") e2 = 20; else if (suf == "Gi") e2 = 30; else if (suf == "Ti") e2 = 40; else if (suf == "Pi") e2 = 50; else if (suf == "Ei") e2 = 60;
    else if (suf == "n") e10 = -9; else if (suf == "u") e10 = -6; else if (suf == "m") e10 = -3; else if (suf == "k") e10 = 3; else if (suf == "M") e10 = 6;
    else if (suf == "G") e10 = 9; else if (suf == "T") e10 = 12; else if (suf == "P") e10 = 15; else if (suf == "E") e10 = 18;
    else if ((suf[0] == 'e' || suf[0] == 'E') && suf.size() > 1) { char* end = nullptr; long v = strtol(suf.c_str() + 1, &end, 10); if (*end || v > 30 || v < -30) { q.ok = false; return q; } e10 = (int)v; }
    else { q.ok = false; return q; }
    if (e2) q.num *= ((i128)1 << e2);
    if (e10 > 0) { int take = std::min(e10, q.dexp); q.dexp -= take; q.num *= pow10i(e10 - take); } else q.dexp += -e10;
    if (q.dexp > 36) q.ok = false;
    if (neg) q.num = -q.num;
    return q;
}
Qty qty_of(const JV& v) { if (v.t == JV::Str || v.t == JV::Num) return parse_qty(v.s); Qty q; q.ok = v.is_null(); return q; }
Qty qty_add(Qty a, const Qty& b) { int d = std::max(a.dexp, b.dexp); a.num = a.num * pow10i(d - a.dexp) + b.num * pow10i(d - b.dexp); a.dexp = d; a.ok = a.ok && b.ok; return a; }
int64_t ceil_div(i128 n, i128 d) { i128 q = n / d; if (n % d != 0 && n > 0) q++; return (int64_t)q; }
int64_t qty_value(const Qty& q) { return ceil_div(q.num, pow10i(q.dexp)); }
int64_t qty_milli(const Qty& q) { return ceil_div(q.num * 1000, pow10i(q.dexp)); }
bool qty_zero(const Qty& q) { return q.num == 0; }

// ===================================================================================================== time
// metav1.Time / time.RFC3339 → ns since the Unix epoch; 0 when absent (the zero Time sorts before every real timestamp too)
int64_t days_from_civil(int64_t y, unsigned m, unsigned d) {
    y -= m <= 2; const int64_t era = (y >= 0 ? y : y - 399) / 400; const unsigned yoe = (unsigned)(y - era * 400);
    const unsigned doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5 + d - 1; const unsigned doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
    return era * 146097 + (int64_t)doe - 719468;
}
bool parse_rfc3339(const std::string& s, int64_t& out) {
    int Y, M, D, h, m, sec; int n = 0;
    if (sscanf(s.c_str(), "%4d-%2d-%2d%*1[Tt ]%2d:%2d:%2d%n", &Y, &M, &D, &h, &m, &sec, &n) != 6 || n == 0) return false;
    const char* p = s.c_str() + n; int64_t frac = 0;
    if (*p == '.' || *p == ',') { p++; int k = 0; while (*p >= '0' && *p <= '9') { if (k < 9) { frac = frac * 10 + (*p - '0'); k++; } p++; } while (k++ < 9) frac *= 10; }
    int64_t off = 0;
    if (*p == 'Z' || *p == 'z') p++;
    else if (*p == '+' || *p == '-') { int oh, om; if (sscanf(p + 1, "%2d:%2d", &oh, &om) != 2) return false; off = (oh * 3600 + om * 60) * (*p == '-' ? -1 : 1); p += 6; }
    else return false;
    if (*p) return false;
    out = ((days_from_civil(Y, M, D) * 86400 + h * 3600 + m * 60 + sec) - off) * 1000000000LL + frac;
    return true;
}
int64_t time_of(const JV& v) { int64_t t; return v.t == JV::Str && parse_rfc3339(v.s, t) ? t : 0; }
// time.ParseDuration: [-+]?([0-9]*(\.[0-9]*)?[a-z]+)+, units ns us µs ms s m h
bool parse_duration(const std::string& s, int64_t& out) {
    const char* p = s.c_str(); bool neg = false; if (*p == '+' || *p == '-') { neg = *p == '-'; p++; }
    if (!strcmp(p, "0")) { out = 0; return true; }
    if (!*p) return false;
    long double total = 0;
    while (*p) {
        long double v = 0, scale = 1; bool any = false;
        while (*p >= '0' && *p <= '9') { v = v * 10 + (*p - '0'); p++; any = true; }
        if (*p == '.') { p++; while (*p >= '0' && *p <= '9') { scale /= 10; v += (*p - '0') * scale; p++; any = true; } }
        if (!any) return false;
        long double unit;
        if (!strncmp(p, "ns", 2)) { unit = 1; p += 2; } else if (!strncmp(p, "us", 2)) { unit = 1e3L; p += 2; } else if (!strncmp(p, "\xC2\xB5s", 3) || !strncmp(p, "\xCE\xBCs", 3)) { unit = 1e3L; p += 3; }
        else if (!strncmp(p, "ms", 2)) { unit = 1e6L; p += 2; } else if (*p == 's') { unit = 1e9L; p++; } else if (*p == 'm') { unit = 60e9L; p++; } else if (*p == 'h') { unit = 3600e9L; p++; } else return false;
        total += v * unit;
    }
    out = (int64_t)llroundl(neg ? -total : total);
    return true;
}

// ===================================================================================================== helpers
// static partition over the host cores; pods are independent of each other until the signatures are interned
template <class F> void parallel_for(size_t n, F f) {
    unsigned hw = std::thread::hardware_concurrency(); if (const char* e = std::getenv("KAI_INGEST_THREADS")) hw = (unsigned)atoi(e);
    size_t T = std::min<size_t>(std::min<size_t>(hw ? hw : 1, 32), n / 512 + 1);
    if (T <= 1) { for (size_t i = 0; i < n; i++) f(i); return; }
    std::vector<std::thread> th;
    for (size_t t = 0; t < T; t++) th.emplace_back([=, &f] { for (size_t i = n * t / T, e = n * (t + 1) / T; i < e; i++) f(i); });
    for (auto& x : th) x.join();
}
std::vector<uint32_t> rank_strings(const std::vector<std::string>& names) {  // byte-wise ascending like Go's string <; ties keep first-seen order
    const size_t n = names.size();
    std::vector<uint32_t> order(n), rank(n);
    for (size_t i = 0; i < n; i++) order[i] = (uint32_t)i;
    auto less = [&](uint32_t a, uint32_t b) { const int c = names[a].compare(names[b]); return c < 0 || (c == 0 && a < b); };  // a strict total order: any sort gives the stable result
    size_t T = 1;
    if (n >= 131072) { unsigned hw = std::thread::hardware_concurrency(); if (const char* e = std::getenv("KAI_INGEST_THREADS")) hw = (unsigned)atoi(e); T = std::min<size_t>(std::min<size_t>(hw ? hw : 1, 16), n / 65536); }  // (called per job as well, on a handful of names)
    if (T <= 1) std::stable_sort(order.begin(), order.end(), less);  // (merge sort: names that arrive in order — the usual export — cost one pass)
    else {  // sorted runs on the host cores, then pairwise merges (each level in parallel)
        std::vector<size_t> cut(T + 1); for (size_t t = 0; t <= T; t++) cut[t] = n * t / T;
        { std::vector<std::thread> th; for (size_t t = 0; t < T; t++) th.emplace_back([&, t] { std::stable_sort(order.begin() + cut[t], order.begin() + cut[t + 1], less); }); for (auto& x : th) x.join(); }
        std::vector<uint32_t> tmp(n);
        while (cut.size() > 2) {
            std::vector<size_t> next; std::vector<std::thread> th;
            for (size_t t = 0; t + 1 < cut.size(); t += 2) {
                if (t + 2 < cut.size()) { th.emplace_back([&, t] { std::merge(order.begin() + cut[t], order.begin() + cut[t + 1], order.begin() + cut[t + 1], order.begin() + cut[t + 2], tmp.begin() + cut[t], less); }); next.push_back(cut[t]); }
                else { std::copy(order.begin() + cut[t], order.begin() + cut[t + 1], tmp.begin() + cut[t]); next.push_back(cut[t]); }
            }
            for (auto& x : th) x.join();
            next.push_back(n); cut.swap(next); order.swap(tmp);
        }
    }
    for (size_t r = 0; r < n; r++) rank[order[r]] = (uint32_t)r;
    return rank;
}
const JV& label_of(const JV& obj, const char* key) { return obj["metadata"]["labels"][key]; }
bool has_label(const JV& obj, const std::string& key) { const JV& l = obj["metadata"]["labels"]; if (!l.is_obj()) return false; for (auto& kv : l.o) if (kv.first == key) return true; return false; }
std::string lower(std::string s) { for (auto& c : s) if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a'); return s; }
bool starts_with(const std::string& s, const char* pre) { return !s.compare(0, strlen(pre), pre); }
// canonical serialisation (object keys sorted) — only used to compare constraint sub-trees for equality
void canon(const JV& v, std::string& out) {
    switch (v.t) {
        case JV::Null: out += "null"; break; case JV::Bool: out += v.b ? "true" : "false"; break; case JV::Num: out += v.s; break;
        case JV::Str: out += '"'; for (char c : v.s) { if (c == '"' || c == '\\') out += '\\'; out += c; } out += '"'; break;
        case JV::Arr: out += '['; for (auto& x : v.a) { canon(x, out); out += ','; } out += ']'; break;
        case JV::Obj: { std::vector<const std::pair<std::string, JV>*> ks; for (auto& kv : v.o) ks.push_back(&kv);
            std::stable_sort(ks.begin(), ks.end(), [](auto* a, auto* b) { return a->first < b->first; });
            out += '{'; for (auto* kv : ks) { out += '"'; out += kv->first; out += "\":"; canon(kv->second, out); out += ','; } out += '}'; break; }
    }
}

// resource names (api/resource_info/resource_requirment.go:17-18,45-71, resource_info.go:53-79, k8s_internal/kubernetes_helpers.go:12-15)
bool is_gpu_name(const std::string& n) { return n == "nvidia.com/gpu" || n == "amd.com/gpu"; }
bool is_mig_name(const std::string& n) { return starts_with(n, "nvidia.com/mig-"); }
// ^nvidia.com/mig-(\d+)g\.(\d+)gb$ (api/common_info/resources/mig.go:13-33): GPU weight and memory of the profile; false when the name does not parse
bool parse_mig_name(const std::string& n, int& g, int64_t& m) {
    const std::string pre = "nvidia.com/mig-"; if (!starts_with(n, pre.c_str())) return false;
    size_t i = pre.size(), j = i; while (j < n.size() && isdigit((unsigned char)n[j])) j++;
    if (j == i || j - i > 6 || n.compare(j, 2, "g.") != 0) return false;
    size_t k = j + 2, l = k; while (l < n.size() && isdigit((unsigned char)n[l])) l++;
    if (l == k || l - k > 9 || n.compare(l, std::string::npos, "gb") != 0) return false;
    g = atoi(n.substr(i, j - i).c_str()); m = atoll(n.substr(k, l - k).c_str());
    return true;
}
bool is_scalar_name(const std::string& n) {  // v1helper.IsExtendedResourceName || IsHugePageResourceName || IsPrefixedNativeResource || IsAttachableVolumeResourceName
    bool native = n.find('/') == std::string::npos || n.find("kubernetes.io/") != std::string::npos;
    if (!native && !starts_with(n, "requests.")) return true;
    return starts_with(n, "hugepages-") || n.find("kubernetes.io/") != std::string::npos || starts_with(n, "attachable-volumes-");
}
bool is_storage_name(const std::string& n) { return n == "ephemeral-storage" || n == "storage"; }

struct Req {  // ResourceRequirements restricted to what the device path carries
    double cpu = 0, mem = 0, gpu = 0; std::map<std::string, int64_t> scalars; bool mig = false, bad = false;
};
// RequirementsFromResourceList (resource_requirment.go:45-71) over an exact sum of Quantities per resource name
Req req_from_list(const std::map<std::string, Qty>& rl) {
    Req r;
    for (auto& kv : rl) {
        const std::string& n = kv.first; const Qty& q = kv.second; if (!q.ok) r.bad = true;
        if (n == "cpu") r.cpu += (double)qty_milli(q);
        else if (n == "memory") r.mem += (double)qty_value(q);
        else if (is_gpu_name(n)) { int64_t v = qty_value(q); if (v >= 1) r.gpu += (double)v; }  // Value() is an integer: below 1 means 0 devices
        else if (is_mig_name(n)) { r.mig = true; r.scalars[n] += qty_value(q); }  // instances (Value): a resource row of its own, described by res_mig_* (ABI v5)
        else if (is_scalar_name(n)) r.scalars[n] += qty_milli(q);
        else if (is_storage_name(n)) r.scalars[n] += qty_value(q);
    }
    return r;
}
std::map<std::string, Qty> list_of(const JV& rl) { std::map<std::string, Qty> m; if (rl.is_obj()) for (auto& kv : rl.o) m[kv.first] = qty_of(kv.second); return m; }

// ---- upstream static Filters (k8s.io/kubernetes v1.34.2; restated from the published algorithm)
struct NodeView { const JV* node; std::map<std::string, std::string> labels; std::string name; };
// component-helpers/scheduling/corev1/nodeaffinity: one NodeSelectorRequirement against labels (or metadata.name for matchFields)
bool match_expr(const JV& ex, const std::map<std::string, std::string>& labels, bool& invalid) {
    const std::string& key = ex["key"].str(); const std::string& op = ex["operator"].str(); const JV& vals = ex["values"];
    auto it = labels.find(key); bool has = it != labels.end();
    auto in_vals = [&]() { if (vals.is_arr()) for (auto& v : vals.a) if (v.str() == it->second) return true; return false; };
    if (op == "In") { if (!vals.is_arr() || vals.a.empty()) { invalid = true; return false; } return has && in_vals(); }
    if (op == "NotIn") { if (!vals.is_arr() || vals.a.empty()) { invalid = true; return false; } return !has || !in_vals(); }
    if (op == "Exists") return has;
    if (op == "DoesNotExist") return !has;
    if (op == "Gt" || op == "Lt") {
        if (!vals.is_arr() || vals.a.size() != 1) { invalid = true; return false; }
        char* e1 = nullptr; long long rhs = strtoll(vals.a[0].str().c_str(), &e1, 10); if (vals.a[0].str().empty() || *e1) { invalid = true; return false; }
        if (!has) return false;
        char* e2 = nullptr; long long lhs = strtoll(it->second.c_str(), &e2, 10); if (it->second.empty() || *e2) return false;
        return op == "Gt" ? lhs > rhs : lhs < rhs;
    }
    invalid = true; return false;
}
bool match_term(const JV& term, const NodeView& nv) {  // a term without any requirement matches nothing; an invalid term matches nothing
    const JV& me = term["matchExpressions"]; const JV& mf = term["matchFields"];
    size_t n = (me.is_arr() ? me.a.size() : 0) + (mf.is_arr() ? mf.a.size() : 0);
    if (n == 0) return false;
    bool invalid = false, all = true;
    if (me.is_arr()) for (auto& ex : me.a) if (!match_expr(ex, nv.labels, invalid)) all = false;
    if (mf.is_arr()) { std::map<std::string, std::string> f{{"metadata.name", nv.name}}; for (auto& ex : mf.a) { if (ex["key"].str() != "metadata.name") { invalid = true; continue; } if (!match_expr(ex, f, invalid)) all = false; } }
    return all && !invalid;
}
// nodeaffinity.GetRequiredNodeAffinity(pod).Match(node): spec.nodeSelector AND required node-affinity terms (ORed)
bool node_affinity_fits(const JV& pod, const NodeView& nv) {
    const JV& sel = pod["spec"]["nodeSelector"];
    if (sel.is_obj()) for (auto& kv : sel.o) { auto it = nv.labels.find(kv.first); if (it == nv.labels.end() || it->second != kv.second.str()) return false; }
    const JV& reqd = pod["spec"]["affinity"]["nodeAffinity"]["requiredDuringSchedulingIgnoredDuringExecution"];
    if (reqd.is_obj()) {
        const JV& terms = reqd["nodeSelectorTerms"]; bool any = false;
        if (terms.is_arr()) for (auto& t : terms.a) if (match_term(t, nv)) { any = true; break; }
        if (!any) return false;
    }
    return true;
}
// tainttoleration Filter: the first NoSchedule / NoExecute taint no toleration tolerates rejects the node (v1helper.ToleratesTaint)
bool tolerates(const JV& tol, const JV& taint) {
    const std::string& te = tol["effect"].str(); if (!te.empty() && te != taint["effect"].str()) return false;
    const std::string& tk = tol["key"].str(); if (!tk.empty() && tk != taint["key"].str()) return false;
    const std::string& op = tol["operator"].str();
    if (op.empty() || op == "Equal") return tol["value"].str() == taint["value"].str();
    return op == "Exists";
}
bool taints_tolerated(const JV& pod, const JV& node) {
    const JV& taints = node["spec"]["taints"]; if (!taints.is_arr()) return true;
    const JV& tols = pod["spec"]["tolerations"];
    for (auto& t : taints.a) {
        const std::string& eff = t["effect"].str(); if (eff != "NoSchedule" && eff != "NoExecute") continue;
        bool ok = false; if (tols.is_arr()) for (auto& tol : tols.a) if (tolerates(tol, t)) { ok = true; break; }
        if (!ok) return false;
    }
    return true;
}

}  // namespace

// ===================================================================================================== the handle
struct kai_ingest {
    kai_snapshot_soa snap{}; kai_config cfg{};
    std::vector<std::string> names[6];
    std::vector<int32_t> actions; std::string bad_action, warnings;
    int R = 4;
    std::vector<double> node_alloc, pod_req, q_deserved, q_limit, q_oqw;
    std::vector<uint32_t> node_flags, node_name_rank, pod_flags, pod_uid_rank, podset_name_rank, job_uid_rank, queue_uid_rank, domain_id_rank, group_name_rank;
    std::vector<int32_t> node_gpu_count, node_class, pod_job, pod_podset, pod_status, pod_node, pod_task_priority, pod_class, pod_nominated,
        podset_job, podset_min, job_queue, job_priority, job_preempt, job_first_pod, job_n_pods, job_first_podset, job_n_podsets, queue_parent, queue_priority,
        topo_level_off, node_domain, domain_level, domain_parent, group_job, group_parent, group_topology, group_req, group_pref, job_root_group,
        podset_group, podset_topology, podset_req, podset_pref;
    std::vector<int64_t> pod_created, job_created, queue_created, job_signature, job_last_start, q_preempt_mrt, q_reclaim_mrt, node_gpu_memory, res_mig_memory;
    std::vector<int32_t> res_mig_gpus;
    std::vector<double> pod_gpu_portion; std::vector<int32_t> pod_gpu_group; std::vector<int64_t> pod_gpu_memory; bool any_fraction = false, any_gpu_memory = false;
    std::vector<uint8_t> class_fit;
    // what the decision writer needs of each pod / job (cache/cache.go:216-330)
    std::vector<std::string> pod_ns, pod_name, pod_uid, job_ns;
    std::vector<double> pod_gpus;
    std::string np_key, np_val;
    void warn(const std::string& m) { if (warnings.size() < 16384) { warnings += m; warnings += '\n'; } }
    int build(const JV& root, const kai_ingest_options* opt, const PodSpans* lazy_pods = nullptr);
};

int kai_ingest::build(const JV& root, const kai_ingest_options* opt, const PodSpans* lazy_pods) {
    if (!root.is_obj()) { g_err = "snapshot.json: top level is not an object"; return KAI_ERR_INVALID_ARG; }
    const JV& conf = root["config"]; const JV& params = root["schedulerParams"]; const JV& raw = root["rawObjects"];
    const bool timing = std::getenv("KAI_INGEST_TIMING") != nullptr;
    auto tnow = [] { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; };
    double tl = tnow(); auto lap = [&](const char* what) { if (timing) { double t = tnow(); std::fprintf(stderr, "kai_ingest:   %-28s %.3f s\n", what, t - tl); tl = t; } };
    int64_t latest_ts = 0; auto seen_ts = [&](int64_t t) { if (t > latest_ts) latest_ts = t; return t; };

    // ---------------------------------------------------------------- configuration (conf/scheduler_conf.go:31-88, conf_util/scheduler_conf_util.go:36-107)
    cfg.abi_version = KAI_ABI_VERSION; cfg.gpu_strategy = KAI_BINPACK; cfg.cpu_strategy = KAI_BINPACK; cfg.k_value = 1.0; cfg.reclaimer_saturation_multiplier = 1.0;
    cfg.restrict_node_scheduling = params["restrictSchedulingNodes"].truthy();
    cfg.max_consolidation_preemptees = (int32_t)params["maxNumberConsolidationPreemptees"].inum();
    cfg.use_scheduling_signatures = params["useSchedulingSignatures"].truthy();
    cfg.allow_consolidating_reclaim = params["allowConsolidatingReclaim"].truthy();
    cfg.full_hierarchy_fairness = params["fullHierarchyFairness"].truthy();
    cfg.min_node_gpu_memory = 100;  // cluster_info.go:242-257 as written: min(100, x) over values x > 100
    std::string scheduler_name = opt && opt->scheduler_name ? opt->scheduler_name : params["schedulerName"].str();
    if (scheduler_name.empty()) scheduler_name = "kai-scheduler";
    static const char* kActionNames[4] = {"allocate", "consolidation", "reclaim", "preempt"};
    for (int i = 0; i < 4; i++) { const JV& d = conf["queueDepthPerAction"][kActionNames[i]]; cfg.queue_depth[i] = d.t == JV::Num ? (int32_t)d.inum() : -1; }
    {
        std::string acts = conf["actions"].str(); if (acts.empty()) acts = "allocate, consolidation, reclaim, preempt, stalegangeviction";
        size_t b = 0;
        while (b <= acts.size()) {
            size_t c = acts.find(',', b); if (c == std::string::npos) c = acts.size();
            std::string a = acts.substr(b, c - b); size_t l = a.find_first_not_of(" \t\n\r"), r = a.find_last_not_of(" \t\n\r"); a = l == std::string::npos ? "" : a.substr(l, r - l + 1);
            int id = -1; for (int i = 0; i < 4; i++) if (a == kActionNames[i]) id = i;
            if (id >= 0) actions.push_back(id); else if (a != "stalegangeviction" && bad_action.empty()) bad_action = a;
            b = c + 1;
        }
    }
    {
        static const std::pair<const char*, uint32_t> kPlugins[] = {{"predicates", KAI_PLUGIN_PREDICATES}, {"proportion", KAI_PLUGIN_PROPORTION}, {"priority", KAI_PLUGIN_PRIORITY},
            {"elastic", KAI_PLUGIN_ELASTIC}, {"nodeavailability", KAI_PLUGIN_NODEAVAILABILITY}, {"resourcetype", KAI_PLUGIN_RESOURCETYPE}, {"subgrouporder", KAI_PLUGIN_SUBGROUPORDER},
            {"taskorder", KAI_PLUGIN_TASKORDER}, {"nominatednode", KAI_PLUGIN_NOMINATEDNODE}, {"nodeplacement", KAI_PLUGIN_NODEPLACEMENT}, {"minruntime", KAI_PLUGIN_MINRUNTIME}, {"topology", KAI_PLUGIN_TOPOLOGY},
            {"gpusharingorder", KAI_PLUGIN_GPUSHARINGORDER}, {"gpupack", KAI_PLUGIN_GPUPACK}, {"gpuspread", KAI_PLUGIN_GPUSPREAD}};
        const JV& tiers = conf["tiers"];
        if (!tiers.is_arr() || tiers.a.empty()) cfg.plugins = KAI_PLUGIN_ALL;  // the default tier list (scheduler_conf_util.go:36-61)
        else for (auto& tier : tiers.a) if (tier["plugins"].is_arr()) for (auto& pl : tier["plugins"].a) {
            const std::string& n = pl["name"].str(); const JV& args = pl["arguments"]; bool known = false;
            for (auto& kp : kPlugins) if (n == kp.first) { cfg.plugins |= kp.second; known = true; }
            if (n == "nodeplacement") { if (args["gpu"].str() == "spread") cfg.gpu_strategy = KAI_SPREAD; if (args["cpu"].str() == "spread") cfg.cpu_strategy = KAI_SPREAD; }  // nodeplacement.go:59-70
            if (n == "proportion") {  // proportion.go:67-93
                if (args["kValue"].t == JV::Str) { char* e = nullptr; double v = strtod(args["kValue"].s.c_str(), &e); if (!*e && !args["kValue"].s.empty()) cfg.k_value = v; }
                if (args["relcaimerSaturationMultiplier"].t == JV::Str) { char* e = nullptr; double v = strtod(args["relcaimerSaturationMultiplier"].s.c_str(), &e); if (!*e && v >= 1.0) cfg.reclaimer_saturation_multiplier = v; }
            }
            if (n == "minruntime") {  // minruntime.go:40-70
                int64_t d;
                if (args["defaultPreemptMinRuntime"].t == JV::Str && parse_duration(args["defaultPreemptMinRuntime"].s, d) && d >= 0) cfg.default_preempt_min_runtime_ns = d;
                if (args["defaultReclaimMinRuntime"].t == JV::Str && parse_duration(args["defaultReclaimMinRuntime"].s, d) && d >= 0) cfg.default_reclaim_min_runtime_ns = d;
                cfg.reclaim_resolve_method = args["reclaimResolveMethod"].str() == "queue" ? 1 : 0;
            }
            if (!known && n != "kubeflow" && n != "ray" && n != "snapshot") warn("plugin '" + n + "' is not on the path (DRA / pod-affinity scoring): ignored");
        }
    }

    // partition selector (conf/scheduler_conf.go:95-112): applied to nodes, queues and pod groups by the lister
    const std::string np_key = params["partitionParams"]["NodePoolLabelKey"].str(), np_val = params["partitionParams"]["NodePoolLabelValue"].str();
    this->np_key = np_key; this->np_val = np_val;
    auto in_partition = [&](const JV& obj) { if (np_key.empty()) return true; if (np_val.empty()) return !has_label(obj, np_key); return label_of(obj, np_key.c_str()).str() == np_val && has_label(obj, np_key); };

    // ---------------------------------------------------------------- nodes (cluster_info.go:229-257, node_info.go:105-156, scheduler_utils.go:12-40)
    std::vector<NodeView> nodes; std::vector<Req> node_res;
    if (raw["nodes"].is_arr()) for (auto& n : raw["nodes"].a) {
        if (!n.is_obj() || !in_partition(n)) continue;
        if (cfg.restrict_node_scheduling && !has_label(n, "node-role.kubernetes.io/gpu-worker") && !has_label(n, "node-role.kubernetes.io/cpu-worker")) continue;  // filterUnmarkedNodes :533-549
        NodeView nv; nv.node = &n; nv.name = n["metadata"]["name"].str();
        const JV& l = n["metadata"]["labels"]; if (l.is_obj()) for (auto& kv : l.o) nv.labels[kv.first] = kv.second.str();
        nodes.push_back(std::move(nv));
    }
    const int N = (int)nodes.size();
    std::map<std::string, int> node_idx; for (int i = 0; i < N; i++) { if (node_idx.count(nodes[i].name)) warn("duplicate node name " + nodes[i].name); node_idx[nodes[i].name] = i; names[KAI_NAME_NODE].push_back(nodes[i].name); }
    node_flags.assign(N, 0); node_gpu_count.assign(N, -1);
    std::vector<std::map<std::string, int64_t>> node_scalars(N); std::vector<double> ncpu(N), nmem(N), ngpu(N), npods(N);
    for (int i = 0; i < N; i++) {
        const JV& n = *nodes[i].node; uint32_t f = 0;
        // ResourceFromResourceList (resource_info.go:53-79): zero quantities are skipped; pods → Value, MIG → Value, storage → Value, other scalars → MilliValue
        const JV& al = n["status"]["allocatable"]; bool mig_res = false;
        if (al.is_obj()) for (auto& kv : al.o) {
            Qty q = qty_of(kv.second); if (!q.ok) { g_err = "node " + nodes[i].name + ": bad quantity for " + kv.first; return KAI_ERR_INVALID_ARG; }
            if (qty_zero(q)) continue;
            const std::string& rn = kv.first;
            if (rn == "cpu") ncpu[i] += (double)qty_milli(q); else if (rn == "memory") nmem[i] += (double)qty_value(q); else if (is_gpu_name(rn)) ngpu[i] += (double)qty_value(q);
            else if (rn == "pods") npods[i] += (double)qty_value(q); else if (is_mig_name(rn)) { mig_res = true; node_scalars[i][rn] += qty_value(q); }
            else if (is_storage_name(rn)) node_scalars[i][rn] += qty_value(q); else if (is_scalar_name(rn)) node_scalars[i][rn] += qty_milli(q);
        }
        if (n["spec"]["unschedulable"].truthy()) f |= KAI_NODE_NOT_READY;
        const JV& conds = n["status"]["conditions"];
        if (conds.is_arr()) for (auto& c : conds.a) {
            const std::string& ty = c["type"].str(); const std::string& st = c["status"].str();
            if (ty == "Ready") { if (st != "True") f |= KAI_NODE_NOT_READY; }
            else if (ty == "MemoryPressure" || ty == "DiskPressure" || ty == "PIDPressure" || ty == "NetworkUnavailable") { if (st != "False") f |= KAI_NODE_NOT_READY; }
        }
        {   // IsMIGEnabled (node_info.go:704-718): the label decides when present (strconv.ParseBool), else any MIG resource
            auto it = nodes[i].labels.find("node-role.kubernetes.io/mig-enabled"); bool mig;
            if (it != nodes[i].labels.end()) { const std::string& v = it->second; mig = v == "1" || v == "t" || v == "T" || v == "true" || v == "TRUE" || v == "True"; } else mig = mig_res;
            if (mig) { f |= KAI_NODE_MIG_ENABLED; auto ms = nodes[i].labels.find("nvidia.com/mig.strategy"); if (ms != nodes[i].labels.end() && ms->second == "mixed") f |= KAI_NODE_MIG_MIXED; if (ms != nodes[i].labels.end() && ms->second == "single") f |= KAI_NODE_MIG_SINGLE; }
        }
        if (nodes[i].labels.count("node-role.kubernetes.io/gpu-worker")) f |= KAI_NODE_GPU_WORKER;
        if (nodes[i].labels.count("node-role.kubernetes.io/cpu-worker")) f |= KAI_NODE_CPU_WORKER;
        { auto it = nodes[i].labels.find("nvidia.com/gpu.count"); if (it != nodes[i].labels.end()) { char* e = nullptr; long v = strtol(it->second.c_str(), &e, 10); if (!it->second.empty() && !*e) node_gpu_count[i] = (int32_t)v; } }  // node_info.go:619-640
        node_flags[i] = f;
        {   // getNodeGpuMemory (node_info.go:673-687): label nvidia.com/gpu.memory in MiB (bytes when >= 1 TiB-in-MiB: gpu-feature-discovery issue 26), floored to a multiple of 100; 100 when absent
            int64_t mem = 100; auto it = nodes[i].labels.find("nvidia.com/gpu.memory");
            if (it != nodes[i].labels.end() && !it->second.empty()) { char* e = nullptr; long long v = strtoll(it->second.c_str(), &e, 10); if (!*e) { if (v >= 1048576) v /= 1048576; mem = v - v % 100; } }
            node_gpu_memory.push_back(mem);
        }
    }
    if (raw["resourceSlices"].is_arr() && !raw["resourceSlices"].a.empty()) warn("resourceSlices present: DRA GPUs are not counted (KAI_NODE_HAS_DRA_GPUS is never set by the ingest)");

    lap("config + nodes");
    // ---------------------------------------------------------------- bind requests (cluster_info.go:328-349, bindrequest_info: key = namespace/podName, failed ones ignored)
    std::map<std::string, std::string> bind_node;
    if (raw["bindRequests"].is_arr()) for (auto& br : raw["bindRequests"].a) {
        const std::string& sel = br["spec"]["selectedNode"].str(); if (!node_idx.count(sel)) continue;
        bool failed = br["status"]["phase"].str() == "Failed" && (br["spec"]["backoffLimit"].t != JV::Num || br["status"]["failedAttempts"].inum() >= br["spec"]["backoffLimit"].inum());
        if (failed) continue;
        bind_node[br["metadata"]["namespace"].str() + "/" + br["spec"]["podName"].str()] = sel;
    }
    // config maps by namespace/name (k8s_internal/predicates/config_maps.go)
    std::set<std::string> config_maps; if (raw["configMaps"].is_arr()) for (auto& cm : raw["configMaps"].a) config_maps.insert(cm["metadata"]["namespace"].str() + "/" + cm["metadata"]["name"].str());

    // ---------------------------------------------------------------- pods (pod_info.go:172-214, 365-445)
    struct PodRec { const JV* pod; int64_t span = -1; std::string ns, name, uid_raw; std::string key, uid, group, subgroup, class_sig, sched_sig; Req req; int32_t status, node, nominated; uint32_t flags; int32_t task_prio; int64_t created; bool unschedulable, placed_anti_affinity; int job = -1, podset = -1; double gpu_portion = 0; int64_t gpu_memory = 0; std::string gpu_group; };
    std::vector<PodRec> pods; std::set<std::string> extra_names;
    bool any_existing_anti_affinity = false;
    // one pod → its record; reads only what was built above (node / bind-request / config-map tables), so pods are converted in parallel
    auto make_pod = [&](const JV& p, PodRec& r) -> std::string {
        r = PodRec{}; r.pod = &p; const JV& md = p["metadata"]; const JV& spec = p["spec"];
        r.ns = md["namespace"].str(); r.name = md["name"].str(); r.uid_raw = md["uid"].str();
        r.key = r.ns + "/" + r.name; r.uid = r.uid_raw; if (r.uid.empty()) r.uid = r.key;
        r.group = md["annotations"]["pod-group-name"].str(); r.subgroup = md["labels"]["kai.scheduler/subgroup-name"].str();
        r.created = time_of(md["creationTimestamp"]);
        // getPodResourceRequest :373-393: exact sum over containers, max with every init container, + overhead (base resources only), pods := 1
        std::map<std::string, Qty> sum;
        if (spec["containers"].is_arr()) for (auto& c : spec["containers"].a) { const JV& rq = c["resources"]["requests"]; if (rq.is_obj()) for (auto& kv : rq.o) { auto it = sum.find(kv.first); Qty q = qty_of(kv.second); if (it == sum.end()) sum[kv.first] = q; else it->second = qty_add(it->second, q); } }
        r.req = req_from_list(sum);
        if (spec["initContainers"].is_arr()) for (auto& c : spec["initContainers"].a) {
            Req ir = req_from_list(list_of(c["resources"]["requests"]));
            r.req.cpu = std::max(r.req.cpu, ir.cpu); r.req.mem = std::max(r.req.mem, ir.mem); r.req.gpu = std::max(r.req.gpu, ir.gpu); r.req.mig |= ir.mig; r.req.bad |= ir.bad;
            for (auto& kv : ir.scalars) { auto it = r.req.scalars.find(kv.first); if (it == r.req.scalars.end() || kv.second > it->second) r.req.scalars[kv.first] = kv.second; }
        }
        if (spec["overhead"].is_obj()) { Req o = req_from_list(list_of(spec["overhead"])); r.req.cpu += o.cpu; r.req.mem += o.mem; for (auto& kv : o.scalars) r.req.scalars[kv.first] += kv.second; }
        if (r.req.bad) return "pod " + r.key + ": bad resource quantity";
        // node + status (getTaskStatus :410-445)
        std::string node_name = spec["nodeName"].str(); auto bit = bind_node.find(r.key); bool has_bind = bit != bind_node.end();
        if (node_name.empty() && has_bind) node_name = bit->second;
        { auto it = node_idx.find(node_name); r.node = it == node_idx.end() ? -1 : it->second; }
        const std::string& phase = p["status"]["phase"].str(); bool deleting = !md["deletionTimestamp"].is_null();
        if (phase == "Running") r.status = deleting ? KAI_POD_RELEASING : KAI_POD_RUNNING;
        else if (phase == "Pending") r.status = deleting ? KAI_POD_RELEASING : !spec["nodeName"].str().empty() ? KAI_POD_BOUND : has_bind ? KAI_POD_BINDING : (spec["schedulingGates"].is_arr() && !spec["schedulingGates"].a.empty()) ? KAI_POD_GATED : KAI_POD_PENDING;
        else if (phase == "Succeeded") r.status = KAI_POD_SUCCEEDED; else if (phase == "Failed") r.status = KAI_POD_FAILED; else r.status = KAI_POD_UNKNOWN;
        { auto it = node_idx.find(p["status"]["nominatedNodeName"].str()); r.nominated = it == node_idx.end() ? -1 : it->second; }
        if (!spec["schedulerName"].str().empty() ? spec["schedulerName"].str() != scheduler_name : true) r.flags |= KAI_POD_FOREIGN_SCHEDULER;  // proportion.go:276-285
        { const JV& tp = md["labels"]["kai.scheduler/task-priority"]; if (tp.t == JV::Str) { char* e = nullptr; long v = strtol(tp.s.c_str(), &e, 10); if (!tp.s.empty() && !*e) { r.flags |= KAI_POD_HAS_TASK_PRIORITY; r.task_prio = (int32_t)v; } } }  // task_order.go:28-63
        // features outside the device path (SURVEY §8b fallback rule)
        const JV& ann = md["annotations"]; bool fb = false;
        for (auto& kv : r.req.scalars) { int g; int64_t m; if (is_mig_name(kv.first) && !parse_mig_name(kv.first, g, m)) fb = true; }  // a MIG profile the device cannot weigh
        if (ann.is_obj()) for (auto& kv : ann.o) if (is_mig_name(kv.first)) {  // updateLegacyMigResourceRequestFromAnnotations (pod_info.go:500-516): the request becomes that profile, the task is legacy
            char* e = nullptr; const std::string& vs = kv.second.str(); long long v = strtoll(vs.c_str(), &e, 10);
            if (vs.empty() || *e) continue;
            for (auto it = r.req.scalars.begin(); it != r.req.scalars.end();) { if (is_mig_name(it->first)) it = r.req.scalars.erase(it); else ++it; }
            r.req.scalars[kv.first] = v; r.req.gpu = 0; r.req.mig = true; r.flags |= KAI_POD_LEGACY_MIG;
        }
        bool gpu_unmodelled = false;
        {   // shared-GPU requests (pod_info.go:463-486): a fraction of ONE device (annotation gpu-fraction; ABI v4 pod_gpu_portion) and MiB of ONE device
            // (annotation gpu-memory; ABI v5 pod_gpu_memory) are described to the device; several devices per pod (gpu-fraction-num-devices), both
            // annotations at once or a value that does not parse leave the pod to the host path
            const std::string& fs = ann["gpu-fraction"].str(); const std::string& ms = ann["gpu-memory"].str(); const bool multi = !ann["gpu-fraction-num-devices"].str().empty();
            char* e = nullptr; const double fv = fs.empty() ? 0.0 : strtod(fs.c_str(), &e); const bool f_ok = !fs.empty() && !*e && fv > 0 && fv < 1;
            char* e2 = nullptr; const long long mv = ms.empty() ? 0 : strtoll(ms.c_str(), &e2, 10); const bool m_ok = !ms.empty() && !*e2 && mv > 0;
            if (f_ok && ms.empty() && !multi) {
                r.gpu_portion = fv; r.req.gpu = (double)std::llround(fv * 100.0) / 100.0;  /* GPUs() is fixed point, 1/100 (gpu_resource_requirment.go:230-234) */ r.gpu_group = md["labels"]["runai-gpu-group"].str();  // common/resources/gpu_sharing.go:87-100
            } else if (m_ok && fs.empty() && !multi) {
                r.gpu_memory = mv; r.req.gpu = 0;  /* NewGpuResourceRequirementWithGpus(0, memory): one device, portion 0 */ r.gpu_group = md["labels"]["runai-gpu-group"].str();
            } else if (!fs.empty() || !ms.empty() || multi) { fb = true; gpu_unmodelled = true; }
        }
        if (spec["resourceClaims"].is_arr() && !spec["resourceClaims"].a.empty()) fb = true;
        if (spec["affinity"]["podAffinity"].is_obj() || spec["affinity"]["podAntiAffinity"].is_obj()) { fb = true; if (spec["affinity"]["podAntiAffinity"]["requiredDuringSchedulingIgnoredDuringExecution"].is_arr() && r.node >= 0) r.placed_anti_affinity = true; }
        if (spec["volumes"].is_arr()) for (auto& v : spec["volumes"].a) if (v["persistentVolumeClaim"].is_obj() || v["ephemeral"].is_obj()) fb = true;
        for (const char* cs : {"containers", "initContainers"}) if (spec[cs].is_arr()) for (auto& c : spec[cs].a) if (c["ports"].is_arr()) for (auto& port : c["ports"].a) if (port["hostPort"].inum() > 0) fb = true;
        if (fb) r.flags |= KAI_POD_CPU_FALLBACK;
        if (gpu_unmodelled || (spec["resourceClaims"].is_arr() && !spec["resourceClaims"].a.empty()))
            r.flags |= KAI_POD_GPU_UNMODELLED;  // GPU state beyond whole devices and one shared device: the device refuses such a pod when it is active
        {   // kai utility pods (api/pod_info/utility_pods.go:13-33, conf/global_config.go:25-26): never "another scheduler's" (proportion.go:276-285);
            // a reservation pod holds its GPU on behalf of the fraction pods of its group, so its own devices are not booked (node_info.go:465)
            const std::string& app = md["labels"]["app"].str();
            const bool reservation = app == "kai-resource-reservation", scaling = app == "scaling-pod";
            if (reservation || scaling) r.flags &= ~(uint32_t)KAI_POD_FOREIGN_SCHEDULER;
            if (reservation) r.req.gpu = 0;
        }
        // ConfigMap pre-filter (config_maps.go): a missing non-optional config map makes the pod unschedulable everywhere
        {
            std::set<std::string> mounted; std::vector<std::string> need;
            bool has_shared = false; std::string shared_cm; if (ann.is_obj()) for (auto& kv : ann.o) if (kv.first == "runai/shared-gpu-configmap") { has_shared = true; shared_cm = kv.second.str(); }
            for (const char* cs : {"containers", "initContainers", "ephemeralContainers"}) if (spec[cs].is_arr()) for (auto& c : spec[cs].a) {
                if (c["volumeMounts"].is_arr()) for (auto& vm : c["volumeMounts"].a) mounted.insert(vm["name"].str());
                if (c["env"].is_arr()) for (auto& ev : c["env"].a) { const JV& ref = ev["valueFrom"]["configMapKeyRef"]; if (ref.is_obj() && !ref["optional"].truthy()) need.push_back(ref["name"].str()); }
                if (c["envFrom"].is_arr()) for (auto& ef : c["envFrom"].a) { const JV& ref = ef["configMapRef"]; if (ref.is_obj() && !ref["optional"].truthy()) need.push_back(ref["name"].str()); }
            }
            if (spec["volumes"].is_arr()) for (auto& v : spec["volumes"].a) {
                if (!mounted.count(v["name"].str())) continue;
                if (v["configMap"].is_obj() && !v["configMap"]["optional"].truthy()) need.push_back(v["configMap"]["name"].str());
                if (v["projected"]["sources"].is_arr()) for (auto& s : v["projected"]["sources"].a) if (s["configMap"].is_obj() && !s["configMap"]["optional"].truthy()) need.push_back(s["configMap"]["name"].str());
            }
            for (auto& cmn : need) { if (has_shared && starts_with(cmn, shared_cm.c_str())) continue; if (!config_maps.count(md["namespace"].str() + "/" + cmn)) r.unschedulable = true; }
        }
        {   // constraint signatures, compared for equality further down: predicate class (node selector, required node affinity, tolerations,
            // schedulability) and the scheduling-constraints signature of api/pod_info/scheduling_constraints_signature.go
            std::string& sig = r.class_sig;
            canon(spec["nodeSelector"], sig); sig += '|'; canon(spec["affinity"]["nodeAffinity"]["requiredDuringSchedulingIgnoredDuringExecution"], sig); sig += '|'; canon(spec["tolerations"], sig); sig += r.unschedulable ? "|U" : "|S";
            std::string& sg = r.sched_sig;
            if (spec["volumes"].is_arr()) for (auto& v : spec["volumes"].a) if (v["persistentVolumeClaim"].is_obj()) { canon(v["persistentVolumeClaim"], sg); }
            sg += '|'; canon(spec["nodeSelector"], sg); sg += '|'; canon(spec["affinity"], sg); sg += '|';
            if (spec["tolerations"].is_arr()) for (auto& t : spec["tolerations"].a) sg += t["key"].str() + "\x03" + t["operator"].str() + "\x03" + t["value"].str() + "\x03" + t["effect"].str() + "\x04";
            sg += '|'; sg += spec["priorityClassName"].str(); sg += '|'; canon(spec["priority"], sg); sg += '|'; canon(spec["topologySpreadConstraints"], sg); sg += '|';
            for (const char* cs : {"containers", "initContainers"}) if (spec[cs].is_arr()) for (auto& c : spec[cs].a) if (c["ports"].is_arr()) for (auto& port : c["ports"].a) { sg += std::to_string(port["hostPort"].inum()); sg += ','; }
        }
        return std::string();
    };
    auto parse_span = [&](int64_t i, JV& out, std::string& perr, size_t& at) -> bool {  // one element of rawObjects.pods into a DOM of its own
        JParser q{lazy_pods->spans[(size_t)i].first, lazy_pods->spans[(size_t)i].second, {}, 0};
        if (!q.value(out)) { perr = q.err.empty() ? "bad value" : q.err; at = (size_t)(q.p - lazy_pods->base); return false; }
        q.ws(); if (q.p != q.e) { perr = "unexpected character"; at = (size_t)(q.p - lazy_pods->base); return false; }
        return true;
    };
    if (lazy_pods && !lazy_pods->spans.empty()) {
        const size_t n = lazy_pods->spans.size();
        pods.resize(n); std::vector<std::string> errs(n), perrs(n); std::vector<size_t> at(n); std::vector<uint8_t> skip(n, 0);
        parallel_for(n, [&](size_t i) {
            JV tmp;  // lives for this one conversion
            if (!parse_span((int64_t)i, tmp, perrs[i], at[i])) return;
            if (!tmp.is_obj()) { skip[i] = 1; return; }
            errs[i] = make_pod(tmp, pods[i]); pods[i].pod = nullptr; pods[i].span = (int64_t)i;
        });
        for (size_t i = 0; i < n; i++) if (!perrs[i].empty()) { g_err = "snapshot.json: " + perrs[i] + " at byte " + std::to_string(at[i]); return KAI_ERR_INVALID_ARG; }
        for (auto& e : errs) if (!e.empty()) { g_err = e; return KAI_ERR_INVALID_ARG; }
        size_t w = 0; for (size_t i = 0; i < n; i++) if (!skip[i]) { if (w != i) pods[w] = std::move(pods[i]); w++; }
        pods.resize(w);
    } else {
        std::vector<const JV*> pv; if (raw["pods"].is_arr()) for (auto& p : raw["pods"].a) if (p.is_obj()) pv.push_back(&p);
        pods.resize(pv.size()); std::vector<std::string> errs(pv.size());
        parallel_for(pv.size(), [&](size_t i) { errs[i] = make_pod(*pv[i], pods[i]); });
        for (auto& e : errs) if (!e.empty()) { g_err = e; return KAI_ERR_INVALID_ARG; }
    }
    {
        for (auto& r : pods) { seen_ts(r.created); if (r.placed_anti_affinity) any_existing_anti_affinity = true; for (auto& kv : r.req.scalars) if (kv.second != 0) extra_names.insert(kv.first); }
    }
    if (any_existing_anti_affinity) { warn("a placed pod carries required pod anti-affinity: every pending pod is routed to the CPU fallback"); for (auto& r : pods) if (r.status == KAI_POD_PENDING) r.flags |= KAI_POD_CPU_FALLBACK; }

    lap("pods");
    // resource columns: cpu, memory, gpu, pods + the scalar resources pods actually request (BaseResource.LessEqual ranges the request's scalars)
    names[KAI_NAME_RESOURCE] = {"cpu", "memory", "gpu", "pods"};
    for (auto& n : extra_names) { if ((int)names[KAI_NAME_RESOURCE].size() < KAI_MAX_RES) names[KAI_NAME_RESOURCE].push_back(n); else { warn("more than " + std::to_string(KAI_MAX_RES - 4) + " scalar resources requested: pods asking for '" + n + "' go to the CPU fallback"); for (auto& r : pods) { auto it = r.req.scalars.find(n); if (it != r.req.scalars.end() && it->second != 0) r.flags |= KAI_POD_CPU_FALLBACK; } } }
    R = (int)names[KAI_NAME_RESOURCE].size();
    node_alloc.assign((size_t)R * N, 0.0);
    for (int i = 0; i < N; i++) {
        node_alloc[(size_t)KAI_RES_CPU * N + i] = ncpu[i]; node_alloc[(size_t)KAI_RES_MEM * N + i] = nmem[i]; node_alloc[(size_t)KAI_RES_GPU * N + i] = ngpu[i]; node_alloc[(size_t)KAI_RES_PODS * N + i] = npods[i];
        for (int k = 4; k < R; k++) { auto it = node_scalars[i].find(names[KAI_NAME_RESOURCE][k]); if (it != node_scalars[i].end()) node_alloc[(size_t)k * N + i] = (double)it->second; }
    }

    // ---------------------------------------------------------------- queues (queue_info.go:45-90, cache/cluster_info/queue.go:22-129)
    struct QRec { std::string name, parent; int32_t prio; int64_t created; double des[3], lim[3], oqw[3]; int64_t pmrt, rmrt; bool alive = true; };
    std::vector<QRec> qs;
    auto queue_rec = [&](const JV& q) {
        QRec r{}; r.name = q["metadata"]["name"].str(); r.parent = q["spec"]["parentQueue"].str();
        r.prio = q["spec"]["priority"].t == JV::Num ? (int32_t)q["spec"]["priority"].inum() : 100;  // DefaultQueuePriority
        r.created = seen_ts(time_of(q["metadata"]["creationTimestamp"]));
        static const char* kRes[3] = {"cpu", "memory", "gpu"};  // KAI_Q_* order
        for (int k = 0; k < 3; k++) { const JV& x = q["spec"]["resources"][kRes[k]]; r.des[k] = x["quota"].num(); r.lim[k] = x["limit"].num(); r.oqw[k] = x["overQuotaWeight"].num(); }
        int64_t d; r.pmrt = q["spec"]["preemptMinRuntime"].t == JV::Str && parse_duration(q["spec"]["preemptMinRuntime"].s, d) ? d : -1;
        r.rmrt = q["spec"]["reclaimMinRuntime"].t == JV::Str && parse_duration(q["spec"]["reclaimMinRuntime"].s, d) ? d : -1;
        return r;
    };
    if (raw["queues"].is_arr()) {
        if (cfg.full_hierarchy_fairness) { for (auto& q : raw["queues"].a) if (q.is_obj() && in_partition(q)) qs.push_back(queue_rec(q)); }
        else {  // ProjectLevelFairness (queue.go:63-74): one synthetic unlimited parent "default"; only queues that had a parent survive, re-parented
            QRec d{}; d.name = "default"; d.prio = 100; for (int k = 0; k < 3; k++) { d.des[k] = -1; d.lim[k] = -1; d.oqw[k] = 1; } d.pmrt = d.rmrt = -1; qs.push_back(d);
            for (auto& q : raw["queues"].a) if (q.is_obj() && in_partition(q) && !q["spec"]["parentQueue"].str().empty()) { QRec r = queue_rec(q); if (r.name == "default") continue; r.parent = "default"; qs.push_back(r); }
        }
    }
    {   // keyed by name (a later duplicate replaces an earlier one, like the map); orphans are deleted with their subtree (queue.go:105-129)
        std::map<std::string, int> last; for (int i = 0; i < (int)qs.size(); i++) { auto it = last.find(qs[i].name); if (it != last.end()) { qs[it->second].alive = false; warn("duplicate queue " + qs[i].name); } last[qs[i].name] = i; }
        bool changed = true;
        while (changed) { changed = false; for (auto& q : qs) if (q.alive && !q.parent.empty()) { auto it = last.find(q.parent); if (it == last.end() || !qs[it->second].alive) { q.alive = false; changed = true; warn("orphan queue " + q.name + " dropped (missing parent " + q.parent + ")"); } } }
        std::vector<QRec> keep; for (auto& q : qs) if (q.alive) keep.push_back(q); qs.swap(keep);
    }
    const int Q = (int)qs.size(); std::map<std::string, int> queue_idx; for (int i = 0; i < Q; i++) { queue_idx[qs[i].name] = i; names[KAI_NAME_QUEUE].push_back(qs[i].name); }
    queue_parent.resize(Q); queue_priority.resize(Q); queue_created.resize(Q); q_deserved.resize(3 * (size_t)Q); q_limit.resize(3 * (size_t)Q); q_oqw.resize(3 * (size_t)Q); q_preempt_mrt.resize(Q); q_reclaim_mrt.resize(Q);
    for (int i = 0; i < Q; i++) {
        queue_parent[i] = qs[i].parent.empty() ? -1 : queue_idx[qs[i].parent]; queue_priority[i] = qs[i].prio; queue_created[i] = qs[i].created; q_preempt_mrt[i] = qs[i].pmrt; q_reclaim_mrt[i] = qs[i].rmrt;
        for (int k = 0; k < 3; k++) { q_deserved[(size_t)k * Q + i] = qs[i].des[k]; q_limit[(size_t)k * Q + i] = qs[i].lim[k]; q_oqw[(size_t)k * Q + i] = qs[i].oqw[k]; }
    }

    // ---------------------------------------------------------------- topologies (plugins/topology/topology_plugin.go:57-110, topology_structs.go:94-101, common.go:70-77)
    std::vector<std::string> topo_names; std::vector<std::vector<std::string>> topo_levels;
    if (raw["topologies"].is_arr()) for (auto& t : raw["topologies"].a) { topo_names.push_back(t["metadata"]["name"].str()); topo_levels.emplace_back(); if (t["spec"]["levels"].is_arr()) for (auto& l : t["spec"]["levels"].a) topo_levels.back().push_back(l["nodeLabel"].str()); }
    const int T = (int)topo_names.size();
    topo_level_off.assign(1, 0); for (int t = 0; t < T; t++) topo_level_off.push_back(topo_level_off.back() + (int)topo_levels[t].size());
    const int TL = topo_level_off.back();
    node_domain.assign((size_t)std::max(TL, 1) * std::max(N, 1), -1);
    std::vector<std::pair<int, std::string>> domain_ids;
    for (int t = 0; t < T; t++) {
        std::map<std::pair<int, std::string>, int> ids;
        for (int i = 0; i < N; i++) {
            bool part = true; for (auto& lv : topo_levels[t]) if (!nodes[i].labels.count(lv)) part = false;
            if (!part) continue;
            int parent = -1; std::string did;
            for (int l = 0; l < (int)topo_levels[t].size(); l++) {
                if (l) { did += "."; }
                did += nodes[i].labels[topo_levels[t][l]];
                auto key = std::make_pair(l, did); auto it = ids.find(key);
                if (it == ids.end()) { it = ids.emplace(key, (int)domain_level.size()).first; domain_level.push_back(topo_level_off[t] + l); domain_parent.push_back(parent); domain_ids.emplace_back(t, did); }
                node_domain[(size_t)(topo_level_off[t] + l) * N + i] = it->second; parent = it->second;
            }
        }
    }
    const int D = (int)domain_level.size(); domain_id_rank.assign(D, 0);
    for (int t = 0; t < T; t++) { std::vector<int> idx; std::vector<std::string> ss; for (int d = 0; d < D; d++) if (domain_ids[d].first == t) { idx.push_back(d); ss.push_back(domain_ids[d].second); } auto rk = rank_strings(ss); for (size_t k = 0; k < idx.size(); k++) domain_id_rank[idx[k]] = rk[k]; }
    struct TC { int topo = -1, req = -1, pref = -1; std::string sig; };
    auto constraint = [&](const JV& tc) {  // api/topology_info.TopologyConstraintInfo → (topology | -1 none | -2 missing, level indices; an unknown level name lies beyond the last level)
        TC c; if (!tc.is_obj() || tc["topology"].str().empty()) return c;
        const std::string& tn = tc["topology"].str(); c.sig = tn + "/" + tc["requiredTopologyLevel"].str() + "/" + tc["preferredTopologyLevel"].str();
        int t = -1; for (int i = 0; i < T; i++) if (topo_names[i] == tn) t = i;
        if (t < 0) { c.topo = -2; return c; }
        auto lv = [&](const std::string& n) { if (n.empty()) return -1; for (int l = 0; l < (int)topo_levels[t].size(); l++) if (topo_levels[t][l] == n) return l; return 1000000; };
        c.topo = t; c.req = lv(tc["requiredTopologyLevel"].str()); c.pref = lv(tc["preferredTopologyLevel"].str()); return c;
    };

    lap("queues + topologies");
    // ---------------------------------------------------------------- pod groups (cluster_info.go:351-400, 495-542; job_info.go:160-251; subgroup_info/factory.go:16-135)
    int32_t default_priority = 50;  // DefaultPodGroupPriority; the first globalDefault PriorityClass overrides it
    std::map<std::string, int32_t> pc_value;
    if (raw["priorityClasses"].is_arr()) { bool found = false; for (auto& pc : raw["priorityClasses"].a) { pc_value[pc["metadata"]["name"].str()] = (int32_t)pc["value"].inum(); if (!found && pc["globalDefault"].truthy()) { default_priority = (int32_t)pc["value"].inum(); found = true; } } }
    std::unordered_map<std::string, std::vector<int>> pods_by_group; pods_by_group.reserve(pods.size() / 2 + 16);  // (only looked up by name below: no order needed)
    for (int i = 0; i < (int)pods.size(); i++) if (!pods[i].group.empty()) pods_by_group[pods[i].group].push_back(i);
    struct PSRec { std::string name; int32_t min; TC tc; int parent_group; };
    std::vector<int> pod_order; std::vector<std::string> job_uids; std::vector<TC> group_tc_v, podset_tc_v; std::vector<std::string> group_names_v;
    std::vector<std::vector<int>> job_podset_ids;
    if (raw["podGroups"].is_arr()) for (auto& pg : raw["podGroups"].a) {
        if (!pg.is_obj() || !in_partition(pg)) continue;
        const std::string pg_name = pg["metadata"]["name"].str(); const JV& spec = pg["spec"];
        const int j = (int)job_queue.size();
        auto qit = queue_idx.find(spec["queue"].str()); job_queue.push_back(qit == queue_idx.end() ? -1 : qit->second);
        int32_t prio = 0; bool preempt = false;
        if (qit != queue_idx.end()) {  // priority / preemptibility are only set when the queue exists (:374-381)
            auto pc = pc_value.find(spec["priorityClassName"].str()); prio = pc == pc_value.end() ? default_priority : pc->second;
            const std::string& pre = spec["preemptibility"].str(); preempt = pre == "preemptible" ? true : pre == "non-preemptible" ? false : prio < 100;  // pkg/common/podgroup/preemptible.go:10-26
        } else warn("pod group " + pg_name + ": queue '" + spec["queue"].str() + "' does not exist");
        job_priority.push_back(prio); job_preempt.push_back(preempt); job_created.push_back(seen_ts(time_of(pg["metadata"]["creationTimestamp"])));
        names[KAI_NAME_JOB].push_back(pg_name); job_uids.push_back(pg_name); job_ns.push_back(pg["metadata"]["namespace"].str());  // PodGroupInfo.UID = PodGroupID(podGroup.Name) (cluster_info.go:379)
        { int64_t ls = 0; const std::string& a = pg["metadata"]["annotations"]["kai.scheduler/last-start-timestamp"].str(); if (!a.empty() && parse_rfc3339(a, ls)) seen_ts(ls); else ls = 0; job_last_start.push_back(ls); }
        // sub-group tree: root = spec.topologyConstraint; entries with children are SubGroupSets, the others PodSets; parents are lower-cased
        const int root_g = (int)group_job.size(); job_root_group.push_back(root_g);
        group_job.push_back(j); group_parent.push_back(-1); group_names_v.push_back(""); group_tc_v.push_back(constraint(spec["topologyConstraint"]));
        std::vector<PSRec> sets; bool tree_ok = true;
        const JV& sgs = spec["subGroups"];
        if (sgs.is_arr() && !sgs.a.empty()) {
            std::map<std::string, const JV*> all; std::map<std::string, std::vector<std::string>> children; std::vector<std::string> order;
            for (auto& sg : sgs.a) { const std::string& n = sg["name"].str(); if (all.count(n)) { tree_ok = false; break; } all[n] = &sg; order.push_back(n); children[sg["parent"].t == JV::Str ? lower(sg["parent"].s) : ""].push_back(n); }
            std::map<std::string, int> set_group{{"", root_g}};
            if (tree_ok) for (auto& n : order) if (children.count(n)) { set_group[n] = (int)group_job.size(); group_job.push_back(j); group_parent.push_back(-1); group_names_v.push_back(n); group_tc_v.push_back(constraint((*all[n])["topologyConstraint"])); }
            if (tree_ok) for (auto& n : order) {
                const JV& sg = *all[n]; std::string par = sg["parent"].t == JV::Str ? lower(sg["parent"].s) : ""; auto pit = set_group.find(par);
                if (pit == set_group.end()) { tree_ok = false; break; }
                if (children.count(n)) group_parent[set_group[n]] = pit->second;
                else sets.push_back(PSRec{n, (int32_t)std::max<int64_t>(sg["minMember"].inum(), 1), constraint(sg["topologyConstraint"]), pit->second});
            }
            if (!tree_ok) { warn("pod group " + pg_name + ": invalid sub-group tree, treated as one default pod-set"); group_job.resize(root_g + 1); group_parent.resize(root_g + 1); group_names_v.resize(root_g + 1); group_tc_v.resize(root_g + 1); sets.clear(); }
        }
        if (sets.empty()) sets.push_back(PSRec{"default", (int32_t)std::max<int64_t>(spec["minMember"].inum(), 1), TC{}, root_g});  // job_info.go:200-216
        job_first_podset.push_back((int32_t)podset_job.size()); job_n_podsets.push_back((int32_t)sets.size());
        std::map<std::string, int> ps_idx; job_podset_ids.emplace_back();
        for (auto& s : sets) { ps_idx[s.name] = (int)podset_job.size(); job_podset_ids.back().push_back((int)podset_job.size()); podset_job.push_back(j); podset_min.push_back(s.min); names[KAI_NAME_PODSET].push_back(s.name); podset_group.push_back(s.parent_group); podset_tc_v.push_back(s.tc); }
        // tasks (AddTaskInfo :231-251): by the pod-group annotation; a pod whose sub-group does not exist is not added to the job
        job_first_pod.push_back((int32_t)pod_order.size()); int cnt = 0;
        auto git = pods_by_group.find(pg_name);
        if (git != pods_by_group.end()) for (int pi : git->second) {
            if (pods[pi].job >= 0) { warn("pod " + pods[pi].key + " matches two pod groups named " + pg_name); continue; }
            auto sit = ps_idx.find(pods[pi].subgroup.empty() ? "default" : pods[pi].subgroup);
            if (sit == ps_idx.end()) { warn("pod " + pods[pi].key + ": sub-group '" + pods[pi].subgroup + "' not found in pod group " + pg_name + "; left out of the job"); continue; }
            pods[pi].job = j; pods[pi].podset = sit->second; pod_order.push_back(pi); cnt++;
        }
        job_n_pods.push_back(cnt);
    }
    const int J = (int)job_queue.size(); const int S = (int)podset_job.size(); const int G = (int)group_job.size();
    for (int i = 0; i < (int)pods.size(); i++) if (pods[i].job < 0) pod_order.push_back(i);  // pods of no job: node accounting only
    const int P = (int)pod_order.size();

    lap("pod groups");
    // ---------------------------------------------------------------- static predicate classes (n4): pods by constraint, nodes by what those constraints can see
    std::vector<int> pclass_of(pods.size(), 0); std::vector<const JV*> pclass_rep; std::vector<bool> pclass_unsched;
    std::deque<JV> pclass_store;  // representatives of a snapshot whose pods were converted from spans: parsed once more, kept
    std::set<std::string> used_keys; bool uses_name_field = false;
    {
        std::unordered_map<std::string, int> ids;
        for (size_t i = 0; i < pods.size(); i++) {
            const std::string& sig = pods[i].class_sig;
            auto it = ids.find(sig);
            if (it == ids.end()) {
                const JV* rep = pods[i].pod;
                if (!rep) { pclass_store.emplace_back(); std::string pe; size_t at = 0; if (!parse_span(pods[i].span, pclass_store.back(), pe, at)) { g_err = "snapshot.json: " + pe + " at byte " + std::to_string(at); return KAI_ERR_INVALID_ARG; } rep = &pclass_store.back(); }
                const JV& spec = (*rep)["spec"];
                it = ids.emplace(sig, (int)pclass_rep.size()).first; pclass_rep.push_back(rep); pclass_unsched.push_back(pods[i].unschedulable);
                if (spec["nodeSelector"].is_obj()) for (auto& kv : spec["nodeSelector"].o) used_keys.insert(kv.first);
                const JV& terms = spec["affinity"]["nodeAffinity"]["requiredDuringSchedulingIgnoredDuringExecution"]["nodeSelectorTerms"];
                if (terms.is_arr()) for (auto& t : terms.a) { if (t["matchExpressions"].is_arr()) for (auto& ex : t["matchExpressions"].a) used_keys.insert(ex["key"].str()); if (t["matchFields"].is_arr() && !t["matchFields"].a.empty()) uses_name_field = true; }
            }
            pclass_of[i] = it->second;
        }
    }
    node_class.assign(N, 0); std::vector<int> nclass_rep;
    {
        std::unordered_map<std::string, int> ids;
        for (int i = 0; i < N; i++) {
            std::string sig; for (auto& k : used_keys) { auto it = nodes[i].labels.find(k); sig += k; sig += it == nodes[i].labels.end() ? "\x01" : "=" + it->second; sig += '\x02'; }
            const JV& taints = (*nodes[i].node)["spec"]["taints"]; if (taints.is_arr()) for (auto& t : taints.a) { const std::string& eff = t["effect"].str(); if (eff == "NoSchedule" || eff == "NoExecute") { sig += t["key"].str() + "\x03" + t["value"].str() + "\x03" + eff + "\x04"; } }
            if (uses_name_field) sig += "\x05" + nodes[i].name;
            auto it = ids.find(sig); if (it == ids.end()) { it = ids.emplace(sig, (int)nclass_rep.size()).first; nclass_rep.push_back(i); }
            node_class[i] = it->second;
        }
    }
    const int PC = std::max<int>(1, (int)pclass_rep.size()), NC = std::max<int>(1, (int)nclass_rep.size());
    class_fit.assign((size_t)PC * NC, 1);
    for (size_t a = 0; a < pclass_rep.size(); a++) for (size_t b = 0; b < nclass_rep.size(); b++) {
        const NodeView& nv = nodes[nclass_rep[b]];
        class_fit[a * NC + b] = !pclass_unsched[a] && node_affinity_fits(*pclass_rep[a], nv) && taints_tolerated(*pclass_rep[a], *nv.node);
    }

    lap("predicate classes");
    // ---------------------------------------------------------------- scheduling-constraints signatures (job_info.go:547-570, podset.go:167-196, scheduling_constraints_signature.go)
    std::vector<int64_t> pod_sig(pods.size(), 0);
    {
        std::unordered_map<std::string, int64_t> ids;
        for (size_t i = 0; i < pods.size(); i++) {
            const std::string& sig = pods[i].sched_sig;
            pod_sig[i] = ids.emplace(sig, (int64_t)ids.size()).first->second;
        }
    }
    job_signature.assign(J, 0);
    {
        std::unordered_map<std::string, int64_t> ids;
        std::vector<std::vector<int64_t>> ps_pods(S);
        for (size_t i = 0; i < pods.size(); i++) if (pods[i].podset >= 0 && !(pods[i].status & (KAI_POD_ALLOCATED | KAI_POD_PIPELINED | KAI_POD_BINDING | KAI_POD_BOUND | KAI_POD_RUNNING))) ps_pods[pods[i].podset].push_back(pod_sig[i]);  // IsActiveAllocatedStatus pods are skipped
        for (int j = 0; j < J; j++) {
            std::vector<std::string> sigs;
            for (int s : job_podset_ids[j]) {
                // a pod-set whose pods are all active-allocated (or that has none) has the EMPTY signature, whatever its constraints (podset.go:150-154 isAllPodsActiveAllocated):
                // it adds nothing to the job's hash (job_info.go:555-569 writes the sorted signatures one after the other)
                if (ps_pods[s].empty()) continue;
                std::string x = podset_tc_v[s].sig; for (int g = podset_group[s]; g >= 0; g = group_parent[g]) { x += '|'; x += group_tc_v[g].sig; }
                std::sort(ps_pods[s].begin(), ps_pods[s].end()); x += '#'; for (int64_t v : ps_pods[s]) { x += std::to_string(v); x += ','; }
                sigs.push_back(std::move(x));
            }
            std::sort(sigs.begin(), sigs.end()); std::string all; for (auto& x : sigs) { all += x; all += ';'; }
            job_signature[j] = ids.emplace(all, (int64_t)ids.size()).first->second;
        }
    }

    lap("signatures");
    // ---------------------------------------------------------------- pods in their final order: pods of a job contiguous, pods of no job last
    pod_req.assign((size_t)R * P, 0.0); pod_job.resize(P); pod_podset.resize(P); pod_status.resize(P); pod_node.resize(P); pod_flags.resize(P); pod_task_priority.resize(P); pod_created.resize(P); pod_class.resize(P); pod_nominated.resize(P);
    std::vector<std::string> uids(P); std::map<std::string, int> gpu_group_ids;
    for (int k = 0; k < P; k++) {
        const PodRec& r = pods[pod_order[k]];
        pod_req[(size_t)KAI_RES_CPU * P + k] = r.req.cpu; pod_req[(size_t)KAI_RES_MEM * P + k] = r.req.mem; pod_req[(size_t)KAI_RES_GPU * P + k] = r.req.gpu; pod_req[(size_t)KAI_RES_PODS * P + k] = 1.0;
        for (int c = 4; c < R; c++) { auto it = r.req.scalars.find(names[KAI_NAME_RESOURCE][c]); if (it != r.req.scalars.end()) pod_req[(size_t)c * P + k] = (double)it->second; }
        pod_job[k] = r.job; pod_podset[k] = r.podset; pod_status[k] = r.status; pod_node[k] = r.node; pod_flags[k] = r.flags; pod_task_priority[k] = r.task_prio; pod_created[k] = r.created;
        pod_class[k] = pclass_of[pod_order[k]]; pod_nominated[k] = r.nominated; uids[k] = r.uid; names[KAI_NAME_POD].push_back(r.key);
        {   // shared-GPU group id: the numeric name itself, or 2^20 + an interned index for any other name (a UUID)
            int32_t gid = -1;
            if (r.gpu_portion > 0 || r.gpu_memory > 0) { any_fraction = true; if (r.gpu_memory > 0) any_gpu_memory = true; if (!r.gpu_group.empty() && (r.status & (KAI_POD_ALLOCATED | KAI_POD_PIPELINED | KAI_POD_BINDING | KAI_POD_BOUND | KAI_POD_RUNNING | KAI_POD_RELEASING))) {
                char* e = nullptr; long v = strtol(r.gpu_group.c_str(), &e, 10);
                if (!*e && v >= 0 && v < (1 << 20)) gid = (int32_t)v; else gid = (1 << 20) + (int32_t)gpu_group_ids.emplace(r.gpu_group, (int)gpu_group_ids.size()).first->second; } }
            pod_gpu_portion.push_back(r.gpu_portion); pod_gpu_group.push_back(gid); pod_gpu_memory.push_back(r.gpu_memory);
        }
        { pod_ns.push_back(r.ns); pod_name.push_back(r.name); pod_uid.push_back(r.uid_raw); pod_gpus.push_back(r.req.gpu); }
    }
    // ranks: the reference's tie-breaks are string compares (framework/session.go:480-485 node name; session_plugins.go:227-260 UID)
    node_name_rank = rank_strings(names[KAI_NAME_NODE]); pod_uid_rank = rank_strings(uids); job_uid_rank = rank_strings(job_uids); queue_uid_rank = rank_strings(names[KAI_NAME_QUEUE]);
    podset_name_rank.assign(S, 0); group_name_rank.assign(G, 0);
    for (int j = 0; j < J; j++) {
        { std::vector<std::string> ss; for (int s : job_podset_ids[j]) ss.push_back(names[KAI_NAME_PODSET][s]); auto rk = rank_strings(ss); for (size_t k = 0; k < ss.size(); k++) podset_name_rank[job_podset_ids[j][k]] = rk[k]; }
        { std::vector<int> gs; std::vector<std::string> ss; for (int g = job_root_group[j]; g < G && group_job[g] == j; g++) { gs.push_back(g); ss.push_back(group_names_v[g]); } auto rk = rank_strings(ss); for (size_t k = 0; k < gs.size(); k++) group_name_rank[gs[k]] = rk[k]; }
    }
    group_topology.resize(G); group_req.resize(G); group_pref.resize(G); for (int g = 0; g < G; g++) { group_topology[g] = group_tc_v[g].topo; group_req[g] = group_tc_v[g].req; group_pref[g] = group_tc_v[g].pref; }
    podset_topology.resize(S); podset_req.resize(S); podset_pref.resize(S); for (int s = 0; s < S; s++) { podset_topology[s] = podset_tc_v[s].topo; podset_req[s] = podset_tc_v[s].req; podset_pref[s] = podset_tc_v[s].pref; }
    cfg.now_ns = opt && opt->now_ns ? opt->now_ns : latest_ts;

    lap("final arrays + ranks");
    // ---------------------------------------------------------------- the struct
    auto ptr = [](auto& v) { typedef typename std::remove_reference<decltype(v)>::type::value_type E; static E dummy{}; return v.empty() ? (const E*)&dummy : (const E*)v.data(); };
    kai_snapshot_soa& s = snap; s.abi_version = KAI_ABI_VERSION; s.n_res = R;
    s.n_nodes = N; s.node_allocatable = ptr(node_alloc); s.node_flags = ptr(node_flags); s.node_gpu_count = ptr(node_gpu_count); s.node_name_rank = ptr(node_name_rank); s.node_class = ptr(node_class);
    s.n_pods = P; s.pod_req = ptr(pod_req); s.pod_job = ptr(pod_job); s.pod_podset = ptr(pod_podset); s.pod_status = ptr(pod_status); s.pod_node = ptr(pod_node); s.pod_flags = ptr(pod_flags);
    s.pod_task_priority = ptr(pod_task_priority); s.pod_created_ns = ptr(pod_created); s.pod_uid_rank = ptr(pod_uid_rank); s.pod_class = ptr(pod_class); s.pod_nominated_node = ptr(pod_nominated);
    s.n_podsets = S; s.podset_job = ptr(podset_job); s.podset_min_available = ptr(podset_min); s.podset_name_rank = ptr(podset_name_rank);
    s.n_jobs = J; s.job_queue = ptr(job_queue); s.job_priority = ptr(job_priority); s.job_preemptible = ptr(job_preempt); s.job_created_ns = ptr(job_created); s.job_uid_rank = ptr(job_uid_rank);
    s.job_first_pod = ptr(job_first_pod); s.job_n_pods = ptr(job_n_pods); s.job_first_podset = ptr(job_first_podset); s.job_n_podsets = ptr(job_n_podsets);
    s.n_queues = Q; s.queue_parent = ptr(queue_parent); s.queue_priority = ptr(queue_priority); s.queue_created_ns = ptr(queue_created); s.queue_uid_rank = ptr(queue_uid_rank);
    s.queue_deserved = ptr(q_deserved); s.queue_limit = ptr(q_limit); s.queue_oqw = ptr(q_oqw); s.queue_usage = nullptr;  // usage lister is nil when a snapshot is replayed (cluster_info.go:158-165)
    s.n_pod_classes = PC; s.n_node_classes = NC; s.class_fit = ptr(class_fit);
    s.n_topologies = T; s.topo_level_off = ptr(topo_level_off); s.n_topo_levels = TL; s.node_domain = ptr(node_domain); s.n_domains = D; s.domain_level = ptr(domain_level); s.domain_parent = ptr(domain_parent); s.domain_id_rank = ptr(domain_id_rank);
    s.n_groups = G; s.group_job = ptr(group_job); s.group_parent = ptr(group_parent); s.group_name_rank = ptr(group_name_rank); s.group_topology = ptr(group_topology); s.group_required_level = ptr(group_req); s.group_preferred_level = ptr(group_pref);
    s.job_root_group = ptr(job_root_group); s.podset_group = ptr(podset_group); s.podset_topology = ptr(podset_topology); s.podset_required_level = ptr(podset_req); s.podset_preferred_level = ptr(podset_pref);
    if (any_fraction) { s.pod_gpu_portion = ptr(pod_gpu_portion); s.pod_gpu_group = ptr(pod_gpu_group); }
    if (any_gpu_memory) s.pod_gpu_memory = ptr(pod_gpu_memory);
    s.node_gpu_memory = ptr(node_gpu_memory);
    {   // MIG profiles among the resource rows (ABI v5)
        bool any = false; res_mig_gpus.assign((size_t)R, 0); res_mig_memory.assign((size_t)R, 0);
        for (int k = 4; k < R; k++) { int g; int64_t m; if (parse_mig_name(names[KAI_NAME_RESOURCE][k], g, m)) { res_mig_gpus[k] = g; res_mig_memory[k] = m; any = true; } }
        if (any) { s.res_mig_gpus = ptr(res_mig_gpus); s.res_mig_memory = ptr(res_mig_memory); }
    }
    s.job_signature = ptr(job_signature); s.job_last_start_ns = ptr(job_last_start); s.queue_preempt_min_runtime_ns = ptr(q_preempt_mrt); s.queue_reclaim_min_runtime_ns = ptr(q_reclaim_mrt);
    return KAI_OK;
}

// ===================================================================================================== zip (APPNOTE 6.3.x: end-of-central-directory → central directory → local header → stored / deflate)
namespace {
uint32_t rd32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t rd16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
bool unzip_member(const std::string& z, const char* member, std::string& out) {
    const unsigned char* b = (const unsigned char*)z.data(); const size_t n = z.size();
    if (n < 22) return false;
    size_t eocd = std::string::npos;
    for (size_t i = n - 22;; i--) { if (rd32(b + i) == 0x06054b50) { eocd = i; break; } if (i == 0 || n - i > 66000) break; }
    if (eocd == std::string::npos) return false;
    size_t cd = rd32(b + eocd + 16); int entries = rd16(b + eocd + 10);
    for (int k = 0; k < entries && cd + 46 <= n; k++) {
        if (rd32(b + cd) != 0x02014b50) return false;
        uint16_t method = rd16(b + cd + 10); uint32_t csize = rd32(b + cd + 20), usize = rd32(b + cd + 24); uint16_t fl = rd16(b + cd + 28), xl = rd16(b + cd + 30), cl = rd16(b + cd + 32); uint32_t lho = rd32(b + cd + 42);
        if (cd + 46 + (size_t)fl + xl + cl > n) return false;  // every offset below in size_t: header fields are untrusted
        std::string name((const char*)b + cd + 46, fl);
        if (name == member) {
            if (csize == 0xFFFFFFFFu || usize == 0xFFFFFFFFu) { g_err = "zip64 members are not supported"; return false; }
            if ((size_t)lho + 30 > n || rd32(b + lho) != 0x04034b50) return false;
            size_t data = (size_t)lho + 30 + rd16(b + lho + 26) + rd16(b + lho + 28);
            if (data > n || (size_t)csize > n - data) return false;
            if (method == 0) { out.assign((const char*)b + data, csize); return true; }
            if (method != 8) { g_err = "unsupported zip compression method"; return false; }
            if ((size_t)usize > ((size_t)1 << 31) || (usize > 1032 && (size_t)usize / 1032 > (size_t)csize + 1)) { g_err = "zip member: implausible uncompressed size"; return false; }  // deflate expands at most 1032:1
            out.resize(usize); z_stream zs{}; if (inflateInit2(&zs, -MAX_WBITS) != Z_OK) return false;
            zs.next_in = (Bytef*)(b + data); zs.avail_in = csize; zs.next_out = (Bytef*)&out[0]; zs.avail_out = usize;
            int rc = inflate(&zs, Z_FINISH); inflateEnd(&zs);
            return rc == Z_STREAM_END && zs.total_out == usize;
        }
        cd += 46 + fl + xl + cl;
    }
    return false;
}
}  // namespace

// ===================================================================================================== C ABI
extern "C" {

int kai_ingest_parse(const char* json, size_t len, const kai_ingest_options* opt, kai_ingest** out) {
    if (!json || !out) { g_err = "null argument"; return KAI_ERR_INVALID_ARG; }
    *out = nullptr; g_err.clear();
    double t_start; { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); t_start = ts.tv_sec + ts.tv_nsec * 1e-9; }
    JV root; std::string perr; size_t perr_at = 0; PodSpans pod_spans;
    if (!parse_document(json, len, root, perr, perr_at, &pod_spans)) { g_err = "snapshot.json: " + perr + " at byte " + std::to_string(perr_at); return KAI_ERR_INVALID_ARG; }
    const bool timing = std::getenv("KAI_INGEST_TIMING") != nullptr;
    auto now = [] { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; };
    double t_parsed = now();
    kai_ingest* h = new kai_ingest();
    int rc = h->build(root, opt, &pod_spans);
    if (timing) std::fprintf(stderr, "kai_ingest: parse %.3f s, build %.3f s (%zu bytes)\n", t_parsed - t_start, now() - t_parsed, len);
    if (rc != KAI_OK) { delete h; return rc; }
    *out = h; return KAI_OK;
}

int kai_ingest_load(const char* path, const kai_ingest_options* opt, kai_ingest** out) {
    if (!path || !out) { g_err = "null argument"; return KAI_ERR_INVALID_ARG; }
    *out = nullptr; g_err.clear();
    FILE* f = fopen(path, "rb"); if (!f) { g_err = std::string("cannot open ") + path; return KAI_ERR_INVALID_ARG; }
    std::string buf; char tmp[1 << 16]; size_t n; while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.append(tmp, n); fclose(f);
    if (buf.size() >= 4 && !memcmp(buf.data(), "PK\x03\x04", 4)) {
        std::string json; if (!unzip_member(buf, "snapshot.json", json)) { if (g_err.empty()) g_err = std::string(path) + ": no readable snapshot.json member"; return KAI_ERR_INVALID_ARG; }
        return kai_ingest_parse(json.data(), json.size(), opt, out);
    }
    return kai_ingest_parse(buf.data(), buf.size(), opt, out);
}

const kai_snapshot_soa* kai_ingest_snapshot(const kai_ingest* h) { return h ? &h->snap : nullptr; }
const kai_config* kai_ingest_config(const kai_ingest* h) { return h ? &h->cfg : nullptr; }
int kai_ingest_actions(const kai_ingest* h, int32_t* out, int cap) {
    if (!h) return KAI_ERR_INVALID_ARG;
    if (!h->bad_action.empty()) { g_err = "failed to find Action " + h->bad_action; return KAI_ERR_INVALID_ARG; }  // conf_util/scheduler_conf_util.go:96-107
    for (int i = 0; i < (int)h->actions.size() && i < cap && out; i++) out[i] = h->actions[i];
    return (int)h->actions.size();
}
const char* kai_ingest_name(const kai_ingest* h, int kind, int idx) { if (!h || kind < 0 || kind > 5 || idx < 0 || idx >= (int)h->names[kind].size()) return nullptr; return h->names[kind][idx].c_str(); }
const char* kai_ingest_warnings(const kai_ingest* h) { return h ? h->warnings.c_str() : ""; }
void kai_ingest_free(kai_ingest* h) { delete h; }
const char* kai_ingest_last_error(void) { return g_err.c_str(); }
// The committed operations as the objects the reference's cache would create for them (cache/cache.go:266-330 createBindRequest for
// Allocate, :216-252 Evict; Pipeline has no cluster side effect, framework/statement.go:197-295) — the data format AFTER the path.
int kai_ingest_decisions_json(const kai_ingest* h, const kai_op* ops, int64_t n_ops, char* out, size_t cap, size_t* len) {
    if (!h || (!ops && n_ops > 0) || !len) return KAI_ERR_INVALID_ARG;
    auto esc = [](const std::string& v) { std::string o = "\""; for (unsigned char c : v) { if (c == '"' || c == '\\') { o += '\\'; o += (char)c; } else if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; } else o += (char)c; } o += '"'; return o; };
    const int P = h->snap.n_pods, N = h->snap.n_nodes, J = h->snap.n_jobs;
    std::string binds, evicts, pipes;
    for (int64_t i = 0; i < n_ops; i++) {
        const kai_op& o = ops[i];
        if (o.pod < 0 || o.pod >= P || ((o.kind == KAI_OP_ALLOCATE || o.kind == KAI_OP_PIPELINE) && (o.node < 0 || o.node >= N))) { g_err = "operation " + std::to_string(i) + " is out of range for this snapshot"; return KAI_ERR_INVALID_ARG; }
        const std::string &ns = h->pod_ns[o.pod], &nm = h->pod_name[o.pod];
        if (o.kind == KAI_OP_ALLOCATE) {
            const std::string& node = h->names[KAI_NAME_NODE][o.node]; double g = h->pod_gpus[o.pod];
            if (!binds.empty()) binds += ',';
            binds += "{\"apiVersion\":\"scheduling.run.ai/v1alpha2\",\"kind\":\"BindRequest\",\"metadata\":{\"name\":" + esc(nm) + ",\"namespace\":" + esc(ns) +
                     ",\"ownerReferences\":[{\"apiVersion\":\"v1\",\"kind\":\"Pod\",\"name\":" + esc(nm) + ",\"uid\":" + esc(h->pod_uid[o.pod]) + "}],\"labels\":{\"selected-node\":" + esc(node);
            if (!h->np_key.empty() && !h->np_val.empty()) binds += "," + esc(h->np_key) + ":" + esc(h->np_val);  // SchedulingNodePoolParams.GetLabels
            binds += "}},\"spec\":{\"podName\":" + esc(nm) + ",\"selectedNode\":" + esc(node) + ",\"receivedResourceType\":\"Regular\",\"receivedGPU\":{";  // node_info.go:746-768 setAcceptedResources
            if (g >= 1) binds += "\"count\":" + std::to_string((long long)g) + ",\"portion\":\"1.00\"}}}"; else binds += "\"portion\":\"0.00\"}}}";
        } else if (o.kind == KAI_OP_EVICT) {
            if (!evicts.empty()) evicts += ',';
            evicts += "{\"namespace\":" + esc(ns) + ",\"name\":" + esc(nm) + ",\"uid\":" + esc(h->pod_uid[o.pod]);
            if (o.job >= 0 && o.job < J) evicts += ",\"podGroup\":{\"namespace\":" + esc(h->job_ns[o.job]) + ",\"name\":" + esc(h->names[KAI_NAME_JOB][o.job]) + "}";
            evicts += "}";
        } else if (o.kind == KAI_OP_PIPELINE) {
            if (!pipes.empty()) pipes += ',';
            pipes += "{\"namespace\":" + esc(ns) + ",\"name\":" + esc(nm) + ",\"node\":" + esc(h->names[KAI_NAME_NODE][o.node]) + "}";
        } else { g_err = "operation " + std::to_string(i) + " has an unknown kind"; return KAI_ERR_INVALID_ARG; }
    }
    std::string doc = "{\"bindRequests\":[" + binds + "],\"evictions\":[" + evicts + "],\"pipelined\":[" + pipes + "]}";
    *len = doc.size();
    if (!out || cap < doc.size() + 1) return KAI_ERR_CAPACITY;
    memcpy(out, doc.c_str(), doc.size() + 1);
    return KAI_OK;
}
int kai_quantity_milli(const char* s, int64_t* out) { if (!s || !out) return KAI_ERR_INVALID_ARG; Qty q = parse_qty(s); if (!q.ok) return KAI_ERR_INVALID_ARG; *out = qty_milli(q); return KAI_OK; }
int kai_quantity_value(const char* s, int64_t* out) { if (!s || !out) return KAI_ERR_INVALID_ARG; Qty q = parse_qty(s); if (!q.ok) return KAI_ERR_INVALID_ARG; *out = qty_value(q); return KAI_OK; }

}  // extern "C"
