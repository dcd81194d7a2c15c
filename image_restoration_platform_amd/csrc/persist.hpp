// persist.hpp -- work distribution of the persistent convolution kernels (conv_rb / conv_w4 / conv_up): which (image, tile,
// n-block, k-chunk) a workgroup runs in its s-th pipeline stage.
//
// Items L = ((img * tiles_per_img + ty * tiles_x + tx) * nblocks + nb) are dealt XCD-aware: the workgroups that share an XCD
// (blockIdx % 8: a label, not an id) walk one contiguous eighth [lo, hi) of the item range, workgroup jx of the group taking
// L = lo + jx, lo + jx + nwx, ...  (halo rows and the n-blocks of a tile are then served by that XCD's L2; speed only).
// A workgroup's stages are consecutive (item k, k-chunk kc), so the cursor advances by additions and carries: the integer
// divisions of a from-scratch decode (4 per stage, ~100 scalar instructions) run once, in the constructor.
#pragma once

namespace ire {

struct PersistItem { int img, ty, tx, nb, tile; };
struct PersistStage { PersistItem it; int kc; };

struct PersistCursor {
    int S;                      // stages this workgroup runs (items x nkc); 0: nothing to do
    int my_items;
    // geometry
    int nkc, nblocks, tiles_x, tiles_y;
    // the stride nwx decomposed: nwx = ((d_img * tiles_y + d_ty) * tiles_x + d_tx) * nblocks + d_nb
    int d_img, d_ty, d_tx, d_nb;
    int s;                      // stage the cursor points at
    PersistStage cur;
    int first_img, last_img;    // images of this workgroup's first and last item (gn_fold.hpp)

    __device__ __forceinline__ PersistCursor(int tiles_x_, int tiles_y_, int nimg, int nblocks_, int nkc_) {
        tiles_x = tiles_x_; tiles_y = tiles_y_; nblocks = nblocks_; nkc = nkc_;
        const int tiles_per_img = tiles_x * tiles_y;
        const int items = tiles_per_img * nimg * nblocks;
        const int G = gridDim.x;
        const int X = G < 8 ? G : 8;
        const int xcd = blockIdx.x % X, jx = blockIdx.x / X;
        const int nwx = (G - xcd + X - 1) / X;                        // workgroups in this XCD group
        const int lo = (int)((long long)items * xcd / X), hi = (int)((long long)items * (xcd + 1) / X);
        my_items = (lo + jx < hi) ? (hi - lo - jx + nwx - 1) / nwx : 0;
        S = my_items * nkc;
        auto split = [&](int L, int& img, int& ty, int& tx, int& nb) {
            nb = L % nblocks;
            const int t = L / nblocks;
            img = t / tiles_per_img;
            const int tile = t - img * tiles_per_img;
            ty = tile / tiles_x;
            tx = tile - ty * tiles_x;
        };
        split(nwx, d_img, d_ty, d_tx, d_nb);
        split(lo + jx, cur.it.img, cur.it.ty, cur.it.tx, cur.it.nb);
        cur.it.tile = cur.it.ty * tiles_x + cur.it.tx;
        cur.kc = 0;
        s = 0;
        first_img = cur.it.img;
        last_img = my_items > 0 ? (lo + jx + (my_items - 1) * nwx) / nblocks / tiles_per_img : first_img;
    }
    // the stage after `cur` (clamped: past the last stage the cursor stays on it, as the pipelines' harmless over-fetch expects)
    __device__ __forceinline__ PersistStage next() {
        if (s + 1 < S) {
            ++s;
            if (++cur.kc == nkc) {
                cur.kc = 0;
                PersistItem& it = cur.it;
                it.nb += d_nb;
                int c = 0;
                if (it.nb >= nblocks) { it.nb -= nblocks; c = 1; }
                it.tx += d_tx + c; c = 0;
                if (it.tx >= tiles_x) { it.tx -= tiles_x; c = 1; }
                it.ty += d_ty + c; c = 0;
                if (it.ty >= tiles_y) { it.ty -= tiles_y; c = 1; }
                it.img += d_img + c;
                it.tile = it.ty * tiles_x + it.tx;
            }
        }
        return cur;
    }
};

}  // namespace ire
