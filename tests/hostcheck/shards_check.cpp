// TEST-ONLY (g++, -DMAS_HOST_EMULATION): the displacement arithmetic of Comm::all_gather_shards (csrc/mas_rccl.hpp: shard_range,
// shard_pieces) for even and uneven shard sizes, without any transport: the pieces of every rank must be the same list, tile
// [0, n_total * per_agent) exactly once in rank order, and the all-gather form must be used exactly when the shards are equal
// (ncclAllGather needs equal counts).  The transports themselves are exercised by tests/test_loopback_ranks.py.
#include <cstdio>
#include <vector>

#include "../../multi_agent_solver_amd/csrc/mas_rccl.hpp"

int main() {
  int bad = 0, cases = 0;
  for (int n_ranks = 1; n_ranks <= 9; ++n_ranks) {
    for (int n_total = n_ranks; n_total <= 70; ++n_total) {
      for (size_t per : {size_t(1), size_t(606), size_t(1606)}) {
        ++cases;
        std::vector<mas::ShardPiece> first;
        size_t covered = 0;
        for (int rank = 0; rank < n_ranks; ++rank) {
          const std::vector<mas::ShardPiece> p = mas::shard_pieces(n_total, n_ranks, rank, per);
          int lo = 0, hi = 0; mas::shard_range(n_total, n_ranks, rank, &lo, &hi);
          if (hi - lo < n_total / n_ranks || hi - lo > n_total / n_ranks + 1) ++bad;           // sizes differ by at most one
          covered += static_cast<size_t>(hi - lo);
          if (n_total % n_ranks == 0) {
            // one in-place all-gather: this rank's own shard at its displacement, equal counts on every rank
            if (p.size() != 1 || p[0].root != -1 || p[0].offset != static_cast<size_t>(lo) * per || p[0].count != static_cast<size_t>(n_total / n_ranks) * per) ++bad;
          } else {
            if (rank == 0) first = p;
            if (p.size() != static_cast<size_t>(n_ranks)) { ++bad; continue; }
            size_t at = 0;
            for (int r = 0; r < n_ranks; ++r) {                                              // same list on every rank, tiling the buffer in rank order
              int rlo = 0, rhi = 0; mas::shard_range(n_total, n_ranks, r, &rlo, &rhi);
              if (p[r].root != r || p[r].offset != at || p[r].count != static_cast<size_t>(rhi - rlo) * per) ++bad;
              if (p[r].offset != first[r].offset || p[r].count != first[r].count) ++bad;
              at += p[r].count;
            }
            if (at != static_cast<size_t>(n_total) * per) ++bad;
          }
        }
        if (covered != static_cast<size_t>(n_total)) ++bad;
      }
    }
  }
  std::printf("shards_check: %d cases, %d mismatches\n", cases, bad);
  return bad == 0 ? 0 : 1;
}
