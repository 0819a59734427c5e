// Tuning knobs of the planner and the launchers: ONE process-wide table instead of ~70 environment variables.
//
// Rounds 1-5 read `getenv("SEFD_<KNOB>")` at 64 sites: a plan was a function of the environment at the moment it was built, and a stray variable
// changed what the library did.  Now every site reads this table through tune_str("<KNOB>") (same contract as getenv: nullptr = not set, else the
// value's text), and the table has exactly two writers:
//   * the single environment variable SEFD_TUNING="KNOB=value,KNOB=value,...", read ONCE, the first time the table is consulted;
//   * the C ABI: sefd_tuning_set(knob, value) (value NULL: unset) and sefd_tuning_clear() (include/sefd.h) - what the tests and the A/B tools call.
// A plan is therefore a function of its sefd_model_config and of this explicit table.  The knobs and their defaults: INTEGRATION.md section 6.
// Knobs that a launcher caches in a function-local static (ring depths, tile thresholds) are read at the first launch that consults them.
#pragma once

namespace sefd {
const char* tune_str(const char* knob);                 // nullptr when the knob is not set; the pointer stays valid until the knob is set again / cleared
void tune_set(const char* knob, const char* value);     // value nullptr: unset
void tune_clear();
}  // namespace sefd
