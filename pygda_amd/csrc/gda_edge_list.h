// Host-side edge list handed across the C ABI by the graph builders (gda_ppmi.cpp,
// gda_smooth.cpp); opaque to callers (include/gda_hip.h: gda_edge_list_size/_fetch/_destroy).
#pragma once
#include <cstdint>
#include <vector>

struct gda_edge_list {
    std::vector<int64_t> src, dst;
    std::vector<float> w;
};
