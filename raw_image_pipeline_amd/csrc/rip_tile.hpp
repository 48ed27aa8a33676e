// rip_tile.hpp -- geometry of the tiled remap, shared by the plan compiler (rip_host.cpp) and the kernels
// (rip_remap.hip): one workgroup gathers a kRemapTileW x kRemapTileH destination tile, four pixels per lane.
#pragma once

// 64 x 16 with 256 lanes measured best on config2: 128 x 16 tiles (512 lanes, -DRIP_TILE_W=128) fetch less but
// run 25-85 % slower -- two or three 8-wave workgroups per CU with a barrier per frame hide latency worse than
// six 4-wave ones.  Round 3, same 256 lanes: 128 x 8 tiles (-DRIP_TILE_W=128 -DRIP_TILE_H=8) 2.59-2.63 ms and 64 x 8
// 2.58 ms against 1.88-1.93 ms per 256 frames on config2 (config5: 2.99 / 3.23 against 2.74-2.78).
#ifndef RIP_TILE_W
#define RIP_TILE_W 64
#endif
#ifndef RIP_TILE_H
#define RIP_TILE_H 16
#endif

namespace rip {
constexpr int kRemapTileW = RIP_TILE_W, kRemapTileH = RIP_TILE_H;
constexpr int kRemapTilePx = kRemapTileW * kRemapTileH;          // plan words per tile
constexpr int kRemapTileThreads = kRemapTilePx / 4;              // lanes per workgroup (4 px each)
constexpr int kRemapGroupsPerRow = kRemapTileW / 4;              // lanes per tile row
static_assert(kRemapTileW == 64 || kRemapTileW == 128, "tile width");
}  // namespace rip
