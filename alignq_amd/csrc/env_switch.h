// env_switch.h — the tuning switches (ALIGNQ_* environment variables) read by the launchers.  They choose between kernel
// geometries, so a value outside the set a launcher was written for must never reach it: anything that is not a whole
// number inside [lo, hi] (or not in the `allowed` list) is reported once on stderr and the built-in default is used.
#pragma once
#include <errno.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>

namespace alignq_env {

inline int env_int(const char* name, int dflt, int lo, int hi) {
  const char* e = getenv(name);
  if (!e || !*e) return dflt;
  char* end = nullptr;
  errno = 0;
  const long v = strtol(e, &end, 10);
  if (errno || end == e || *end != '\0' || v < lo || v > hi) {
    fprintf(stderr, "alignq: %s=\"%s\" is not an integer in [%d, %d]; using %d\n", name, e, lo, hi, dflt);
    return dflt;
  }
  return (int)v;
}

template <int N>
inline int env_choice(const char* name, int dflt, const int (&allowed)[N]) {
  const int v = env_int(name, dflt, INT_MIN, INT_MAX);
  for (int i = 0; i < N; i++)
    if (allowed[i] == v) return v;
  fprintf(stderr, "alignq: %s=%d is not one of the supported values; using %d\n", name, v, dflt);
  return dflt;
}

}  // namespace alignq_env
