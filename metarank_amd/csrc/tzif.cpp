// TZif reader (RFC 8536) + POSIX TZ footer rules: see tzif.hpp.  Host only.
#include "tzif.hpp"

#include <sys/stat.h>

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>

namespace mrk {
namespace {

int64_t be(const uint8_t *p, int n) {
  uint64_t v = 0;
  for (int i = 0; i < n; ++i) v = (v << 8) | p[i];
  if (n == 4) return (int64_t)(int32_t)(uint32_t)v;
  return (int64_t)v;
}

int64_t fdiv(int64_t a, int64_t b) { int64_t q = a / b, r = a % b; return (r != 0 && ((r < 0) != (b < 0))) ? q - 1 : q; }

bool is_leap(int64_t y) { return (y % 4 == 0 && y % 100 != 0) || y % 400 == 0; }

int64_t days_from_civil(int64_t y, int m, int d) {
  y -= m <= 2;
  const int64_t era = fdiv(y, 400);
  const uint64_t yoe = (uint64_t)(y - era * 400);
  const uint64_t doy = (153 * (uint64_t)(m > 2 ? m - 3 : m + 9) + 2) / 5 + (uint64_t)d - 1;
  const uint64_t doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
  return era * 146097 + (int64_t)doe - 719468;
}

int64_t year_of_day(int64_t day) {
  int64_t z = day + 719468;
  const int64_t era = fdiv(z, 146097);
  const uint64_t doe = (uint64_t)(z - era * 146097);
  const uint64_t yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
  const uint64_t doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
  const uint64_t mp = (5 * doy + 2) / 153;
  const int month = (int)(mp < 10 ? mp + 3 : mp - 9);
  return (int64_t)yoe + era * 400 + (month <= 2 ? 1 : 0);
}

// the local second-of-epoch (as if UTC) at which `r` fires in year y
int64_t rule_local(const TzRule &r, int64_t y) {
  int64_t day;
  if (r.kind == 0) {
    static const int mdays[] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
    const int64_t first = days_from_civil(y, r.month, 1);
    const int dow_first = (int)(((first % 7) + 7 + 4) % 7);   // 0 = Sunday; 1970-01-01 = Thursday (4)
    int d = 1 + ((r.day - dow_first) % 7 + 7) % 7 + 7 * (r.week - 1);
    const int len = mdays[r.month - 1] + (r.month == 2 && is_leap(y) ? 1 : 0);
    while (d > len) d -= 7;                                  // week 5 = the last such weekday of the month
    day = first + d - 1;
  } else if (r.kind == 1) {
    int n = r.day;                                           // 1..365, February 29 is never counted
    if (is_leap(y) && n >= 60) n += 1;
    day = days_from_civil(y, 1, 1) + n - 1;
  } else {
    day = days_from_civil(y, 1, 1) + r.day;
  }
  return day * 86400 + r.time;
}

// [+-]hh[:mm[:ss]] -> seconds; advances p
bool parse_hms(const char *&p, int32_t &out, bool allow_sign) {
  int sign = 1;
  if (allow_sign && (*p == '+' || *p == '-')) { sign = *p == '-' ? -1 : 1; ++p; }
  if (!isdigit((unsigned char)*p)) return false;
  long h = 0, m = 0, s = 0;
  while (isdigit((unsigned char)*p)) { h = h * 10 + (*p - '0'); ++p; if (h > 1000) return false; }
  if (*p == ':') {
    ++p;
    if (!isdigit((unsigned char)*p)) return false;
    while (isdigit((unsigned char)*p)) { m = m * 10 + (*p - '0'); ++p; if (m > 1000) return false; }
    if (*p == ':') {
      ++p;
      if (!isdigit((unsigned char)*p)) return false;
      while (isdigit((unsigned char)*p)) { s = s * 10 + (*p - '0'); ++p; if (s > 1000) return false; }
    }
  }
  out = (int32_t)(sign * (h * 3600 + m * 60 + s));
  return true;
}

bool skip_name(const char *&p) {
  if (*p == '<') {
    ++p;
    while (*p && *p != '>') ++p;
    if (*p != '>') return false;
    ++p;
    return true;
  }
  int n = 0;
  while (isalpha((unsigned char)*p)) { ++p; ++n; }
  return n >= 3;
}

bool parse_rule(const char *&p, TzRule &r) {
  if (*p == 'M') {
    ++p;
    r.kind = 0;
    char *e;
    r.month = (int)strtol(p, &e, 10); if (e == p || *e != '.') return false; p = e + 1;
    r.week = (int)strtol(p, &e, 10); if (e == p || *e != '.') return false; p = e + 1;
    r.day = (int)strtol(p, &e, 10); if (e == p) return false; p = e;
    if (r.month < 1 || r.month > 12 || r.week < 1 || r.week > 5 || r.day < 0 || r.day > 6) return false;
  } else if (*p == 'J') {
    ++p;
    char *e;
    r.kind = 1;
    r.day = (int)strtol(p, &e, 10); if (e == p || r.day < 1 || r.day > 365) return false; p = e;
  } else if (isdigit((unsigned char)*p)) {
    char *e;
    r.kind = 2;
    r.day = (int)strtol(p, &e, 10); if (e == p || r.day < 0 || r.day > 365) return false; p = e;
  } else return false;
  r.time = 7200;
  if (*p == '/') { ++p; if (!parse_hms(p, r.time, true)) return false; }
  return true;
}

// "CET-1CEST,M3.5.0,M10.5.0/3": std offset [dst [offset] [,start[/time],end[/time]]].  POSIX offsets are WEST of Greenwich.
bool parse_footer(const std::string &tz, TzRules &out) {
  const char *p = tz.c_str();
  if (!*p) return false;
  if (!skip_name(p)) return false;
  int32_t off;
  if (!parse_hms(p, off, true)) return false;
  out.std_off = -off;
  out.has_dst = false;
  if (*p) {
    if (!skip_name(p)) return false;
    out.dst_off = out.std_off + 3600;
    if (*p && *p != ',') { if (!parse_hms(p, off, true)) return false; out.dst_off = -off; }
    if (*p == ',') {
      ++p;
      if (!parse_rule(p, out.start) || *p != ',') return false;
      ++p;
      if (!parse_rule(p, out.end)) return false;
      out.has_dst = true;
    } else if (*p == 0) {   // a daylight name without rules: POSIX leaves it to the implementation; tzcode applies the US rules
      const char *us = "M3.2.0,M11.1.0";
      const char *q = us;
      if (!parse_rule(q, out.start)) return false;
      ++q;
      if (!parse_rule(q, out.end)) return false;
      out.has_dst = true;
    }
    if (*p != 0) return false;
  }
  out.has_footer = true;
  return true;
}

int32_t footer_offset(const TzRules &z, int64_t t) {
  if (!z.has_dst) return z.std_off;
  // the year the instant falls into (by standard time), then its neighbours' rules are never needed: a daylight period
  // that spans the new year (southern hemisphere) is handled by comparing inside ONE year's pair of instants
  const int64_t y = year_of_day(fdiv(t + z.std_off, 86400));
  for (int64_t yy = y - 1; yy <= y + 1; ++yy) {
    const int64_t s = rule_local(z.start, yy) - z.std_off;   // daylight starts: wall clock is standard time
    const int64_t e = rule_local(z.end, yy) - z.dst_off;     // ... ends: wall clock is daylight time
    if (s < e) { if (t >= s && t < e) return z.dst_off; }
    else { if (t >= s && t < rule_local(z.end, yy + 1) - z.dst_off) return z.dst_off; }
  }
  return z.std_off;
}

}  // namespace

int32_t TzRules::offset_at(int64_t t) const {
  if (trans.empty() || t < trans.front()) return trans.empty() && has_footer ? footer_offset(*this, t) : first;
  if (t >= trans.back() && has_footer) return footer_offset(*this, t);
  const size_t i = (size_t)(std::upper_bound(trans.begin(), trans.end(), t) - trans.begin()) - 1;
  return after[i];
}

bool tz_parse(const uint8_t *b, size_t len, TzRules &out) {
  auto header = [&](size_t at, int64_t (&cnt)[6], int &version) {
    if (at + 44 > len || memcmp(b + at, "TZif", 4) != 0) return false;
    version = b[at + 4] ? b[at + 4] - '0' : 1;
    for (int i = 0; i < 6; ++i) cnt[i] = be(b + at + 20 + 4 * i, 4);   // isutcnt isstdcnt leapcnt timecnt typecnt charcnt
    for (int i = 0; i < 6; ++i) if (cnt[i] < 0 || cnt[i] > (1 << 20)) return false;
    return true;
  };
  int64_t c[6];
  int version = 1;
  if (!header(0, c, version)) return false;
  size_t at = 44;
  int tsz = 4;
  if (version >= 2) {   // skip the 32-bit block, read the 64-bit one
    at += (size_t)(c[3] * 4 + c[3] + c[4] * 6 + c[5] + c[2] * 8 + c[1] + c[0]);
    if (!header(at, c, version)) return false;
    at += 44;
    tsz = 8;
  }
  const int64_t timecnt = c[3], typecnt = c[4], charcnt = c[5], leapcnt = c[2], isstd = c[1], isut = c[0];
  if (typecnt < 1) return false;
  const size_t need = (size_t)(timecnt * tsz + timecnt + typecnt * 6 + charcnt + leapcnt * (tsz + 4) + isstd + isut);
  if (at + need > len) return false;
  const uint8_t *tt = b + at, *idx = tt + timecnt * tsz, *types = idx + timecnt;
  out = TzRules();
  out.trans.resize((size_t)timecnt);
  out.after.resize((size_t)timecnt);
  for (int64_t i = 0; i < timecnt; ++i) {
    out.trans[(size_t)i] = be(tt + i * tsz, tsz);
    const int ti = idx[i];
    if (ti >= typecnt) return false;
    out.after[(size_t)i] = (int32_t)be(types + 6 * ti, 4);
    if (i > 0 && out.trans[(size_t)i] < out.trans[(size_t)i - 1]) return false;
  }
  // before the first transition: the first standard-time type if there is one, else type 0 (tzcode's rule)
  int first_type = 0;
  if (timecnt > 0) {
    for (int ti = 0; ti < typecnt; ++ti)
      if (types[6 * ti + 4] == 0) { first_type = ti; break; }
  }
  out.first = (int32_t)be(types + 6 * first_type, 4);
  at += need;
  if (tsz == 8 && at < len && b[at] == '\n') {
    const uint8_t *e = (const uint8_t *)memchr(b + at + 1, '\n', len - at - 1);
    if (e) {
      const std::string tz((const char *)b + at + 1, (size_t)(e - (b + at + 1)));
      if (!tz.empty() && !parse_footer(tz, out)) out.has_footer = false;
    }
  }
  return true;
}

namespace {
std::mutex g_tz_mu;
std::map<std::string, std::unique_ptr<TzRules>> g_tz_cache;   // nullptr = known to be missing
std::string g_tz_dir;
bool g_tz_dir_known = false;
}  // namespace

TzLookup tz_lookup(const std::string &region, const TzRules **out) {
  std::lock_guard<std::mutex> lk(g_tz_mu);
  if (!g_tz_dir_known) {
    std::vector<std::string> cands;
    if (const char *e = getenv("MRK_TZDIR")) cands.push_back(e);
    if (const char *e = getenv("TZDIR")) cands.push_back(e);
    for (const char *d : {"/usr/share/zoneinfo", "/usr/lib/zoneinfo", "/usr/share/lib/zoneinfo", "/etc/zoneinfo"}) cands.push_back(d);
    for (const std::string &d : cands) {
      struct stat st;
      if (!d.empty() && stat(d.c_str(), &st) == 0 && S_ISDIR(st.st_mode)) { g_tz_dir = d; break; }
    }
    g_tz_dir_known = true;
  }
  if (g_tz_dir.empty()) return TzLookup::NoTzdata;
  // ZoneRegion.checkName: [A-Za-z][A-Za-z0-9~/._+-]+ ; and never out of the directory
  if (region.size() < 2 || region.size() > 64 || !isalpha((unsigned char)region[0]) || region.find("..") != std::string::npos) return TzLookup::UnknownRegion;
  for (char ch : region)
    if (!(isalnum((unsigned char)ch) || ch == '~' || ch == '/' || ch == '.' || ch == '_' || ch == '+' || ch == '-')) return TzLookup::UnknownRegion;
  // files a zoneinfo directory holds that java.time's ZoneRulesProvider does not know as region ids: the posix/ and right/
  // trees (right/ carries leap seconds: offsets tens of seconds off), posixrules, localtime, Factory - ZonedDateTime.parse fails
  // on them in the reference (the feature is then missing), so they are unknown regions here too.  (Which ids exist still follows
  // the HOST's tzdata release, not the JVM's bundled tzdb: a zone renamed between the two releases resolves on one side only.)
  if (region.rfind("posix/", 0) == 0 || region.rfind("right/", 0) == 0 || region == "posixrules" || region == "localtime" || region == "Factory")
    return TzLookup::UnknownRegion;
  auto it = g_tz_cache.find(region);
  if (it == g_tz_cache.end()) {
    std::unique_ptr<TzRules> r;
    std::ifstream f(g_tz_dir + "/" + region, std::ios::binary);
    if (f) {
      std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
      r.reset(new TzRules());
      if (!tz_parse(bytes.data(), bytes.size(), *r)) r.reset();
    }
    it = g_tz_cache.emplace(region, std::move(r)).first;
  }
  if (!it->second) return TzLookup::UnknownRegion;
  *out = it->second.get();
  return TzLookup::Ok;
}

// tests: forget the directory and the cache (the environment changed)
void tz_debug_reset() {
  std::lock_guard<std::mutex> lk(g_tz_mu);
  g_tz_cache.clear();
  g_tz_dir.clear();
  g_tz_dir_known = false;
}

}  // namespace mrk
