// nim_tables.cpp -- host emulation of Nim 1.6 Table iteration order (lib/pure/collections/
// tables.nim + tableimpl.nim + hashcommon.nim): open addressing, linear probing, enlarge at 2/3
// load re-inserting in slot order.  STRling iterates Table[(tid, repeat)] with `mpairs`
// (call.nim:223, merge.nim:172), so this order IS the row order of -bounds.txt.
#include "nim_tables.h"
#include <vector>

namespace nim {

// hcodes: hash of each distinct key, in first-insertion order (keys are distinct).
// Returns the key indices in slot order of the final table.
std::vector<int64_t> table_slot_order(const std::vector<uint64_t> &hcodes, uint64_t initial_size) {
  uint64_t len = slots_needed(initial_size), counter = 0;
  std::vector<uint64_t> hc(len, 0);
  std::vector<int64_t> id(len, -1);
  for (size_t q = 0; q < hcodes.size(); ++q) {
    uint64_t h = hcodes[q] ? hcodes[q] : 314159265ull;   // genHashImpl: 0 is the "empty" marker
    if (must_rehash(len, counter)) {
      const uint64_t nl = len * 2;
      std::vector<uint64_t> nh(nl, 0);
      std::vector<int64_t> ni(nl, -1);
      for (uint64_t i = 0; i < len; ++i)
        if (hc[i]) {
          uint64_t j = hc[i] & (nl - 1);
          while (nh[j]) j = (j + 1) & (nl - 1);
          nh[j] = hc[i];
          ni[j] = id[i];
        }
      hc.swap(nh);
      id.swap(ni);
      len = nl;
    }
    uint64_t j = h & (len - 1);
    while (hc[j]) j = (j + 1) & (len - 1);
    hc[j] = h;
    id[j] = (int64_t)q;
    ++counter;
  }
  std::vector<int64_t> order;
  order.reserve(hcodes.size());
  for (uint64_t i = 0; i < len; ++i)
    if (hc[i]) order.push_back(id[i]);
  return order;
}

}  // namespace nim
