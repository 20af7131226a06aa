// Time-zone rules for ISO date-times that name a region - "2011-12-03T10:15:30+01:00[Europe/Paris]" - which
// LocalDateTimeFeature.scala:43-46 hands to ZonedDateTime.parse(_, ISO_DATE_TIME): java.time resolves the INSTANT from the
// written offset and the LOCAL time from the region's rules at that instant.  The JVM carries its own tzdb; a native host
// reads the system's zoneinfo (TZif, RFC 8536): the 64-bit transition table, then - for instants after the last transition,
// which is all of the present in the "slim" files current tzdata packages ship - the POSIX TZ string of the footer.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace mrk {

struct TzRule {          // one side of a POSIX TZ daylight rule: Mm.w.d[/time] | Jn[/time] | n[/time]
  int kind = 0;          // 0 = M, 1 = J (1..365, no leap day), 2 = zero-based day of year
  int month = 0, week = 0, day = 0;
  int32_t time = 7200;   // seconds after local midnight (may be negative or > 24 h)
};

struct TzRules {
  std::vector<int64_t> trans;   // UTC instants of the transitions, ascending
  std::vector<int32_t> after;   // offset in force from trans[i] on
  int32_t first = 0;            // offset before the first transition
  bool has_footer = false;      // the TZ string: what holds after the last transition
  int32_t std_off = 0, dst_off = 0;
  bool has_dst = false;
  TzRule start, end;            // daylight time begins / ends (local wall clock of the side being left)
  int32_t offset_at(int64_t utc_epoch_second) const;
};

// parses the bytes of a TZif file; false when they are not one
bool tz_parse(const uint8_t *bytes, size_t len, TzRules &out);

enum class TzLookup { Ok, UnknownRegion, NoTzdata };
// rules of `region` ("Europe/Paris") from the zoneinfo directory: $MRK_TZDIR, $TZDIR, /usr/share/zoneinfo, /usr/lib/zoneinfo,
// /usr/share/lib/zoneinfo, /etc/zoneinfo - the first that exists; cached per process.  NoTzdata: no such directory on this host.
TzLookup tz_lookup(const std::string &region, const TzRules **out);
void tz_debug_reset();   // tests: forget the directory and the cache (the environment changed)

}  // namespace mrk
