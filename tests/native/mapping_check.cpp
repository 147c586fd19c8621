// Host-side check of the tile kernel's block -> (query tile, slab) deal (lotus_amd/csrc/lvs_tile.h): every item of a launch
// is visited by exactly one block, the grid has no block past the last group, and the per-XCD round count the planner uses
// equals a direct count over the blocks (XCD = block % 8).  Built and run by tests/test_capi.py with hipcc (host code only).
#include "../../lotus_amd/csrc/lvs_tile.h"
#include <cstdio>
#include <vector>

int main() {
    int cases = 0;
    for (int nqt : {1, 2, 7, 8, 31, 32, 33, 40, 49, 98, 196, 391, 513})
        for (int nslab : {1, 2, 3, 4, 5, 9, 13, 16, 21, 36, 64})
            for (int gq : {1, 2, 4, 8, 16, 32})
                for (int lead = 0; lead <= 1; ++lead) {
                    if (lead > nslab) continue;
                    const LvsTileGroups gr = lvs_tile_groups(nqt, nslab, gq, lead);
                    const int nb = lvs_tile_grid_blocks(nqt, nslab, gq, lead);
                    if (nb != gr.total * 32) { printf("grid %d != %d groups * 32\n", nb, gr.total); return 1; }
                    std::vector<int> seen((size_t)nqt * nslab, 0);
                    long long per_xcd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    for (int b = 0; b < nb; ++b) {
                        int g, r, qt, slab;
                        if (!lvs_tile_block_slot(gr, b, g, r)) { printf("block %d of %d has no group\n", b, nb); return 1; }
                        if (!lvs_tile_group_slot(nqt, nslab, gq, lead, gr, g, r, qt, slab)) continue;
                        if (qt < 0 || qt >= nqt || slab < 0 || slab >= nslab) { printf("item out of range\n"); return 1; }
                        ++seen[(size_t)qt * nslab + slab];
                        ++per_xcd[b & 7];
                    }
                    int lo_b, lo_r;
                    if (lvs_tile_block_slot(gr, nb, lo_b, lo_r)) { printf("block past the grid maps to a group\n"); return 1; }
                    for (size_t i = 0; i < seen.size(); ++i)
                        if (seen[i] != 1) {
                            printf("nqt %d nslab %d gq %d lead %d: item (%zu, %zu) visited %d times\n", nqt, nslab, gq, lead,
                                   i / nslab, i % nslab, seen[i]);
                            return 1;
                        }
                    long long worst = 0;
                    for (int x = 0; x < 8; ++x) worst = per_xcd[x] > worst ? per_xcd[x] : worst;
                    const int rounds = (int)((worst + 31) / 32);
                    if (rounds != lvs_tile_xcd_rounds(nqt, nslab, gq, lead)) {
                        printf("nqt %d nslab %d gq %d lead %d: rounds %d, planner says %d\n", nqt, nslab, gq, lead, rounds,
                               lvs_tile_xcd_rounds(nqt, nslab, gq, lead));
                        return 1;
                    }
                    // the deal is balanced to within one 32-slot group plus one item per remainder group
                    ++cases;
                }
    printf("ok %d cases\n", cases);
    return 0;
}
