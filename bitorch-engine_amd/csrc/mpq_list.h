// Device-resident entry of a layer-list launch (bie_mpq_list_*): shared by the lookup/FMA list kernel (mpq_list.hip, M <= 2) and the
// lookup/matrix-pipe list kernel (mpq_gemv_lut.hip, 3 <= M <= 16).
#pragma once
#include "bie_common.h"

namespace bie {

struct ListEntry {          // 128 bytes, read with scalar loads
    const uint32_t* qw;
    const uint16_t* scales;
    const void* zeros;
    const uint16_t* bias;
    const uint16_t* x;
    uint16_t* y;
    unsigned long long* gran;  // [S-1][M][tiles*64] {fp32 partial, tag} granules of this entry (NULL when S == 1)
    unsigned* gen;             // [tiles] generation words of this entry's column tiles
    unsigned* done;            // completion counter of this entry (tiles finished, monotonic over launches)
    const unsigned* dep_done;  // the producer's counter (NULL: independent)
    int N, K, G, gpw, S, hshift, tiles, dep_tiles;
    unsigned qw_bytes, sc_bytes, ze_bytes, pad0;
};
static_assert(sizeof(ListEntry) == 128, "ListEntry layout");

}  // namespace bie
