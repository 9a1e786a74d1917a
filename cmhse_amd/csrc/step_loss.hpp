// step_loss.hpp — workspace layout shared by the two halves of cmhse_step_losses_fwd / _bwd
// (sim.hip holds the forward, bwd.hip the backward).  Everything a training step's contrastive
// terms need between F.normalize of the encoder outputs and the gradients wrt those outputs lives
// in ONE caller-provided workspace, so the step runs the whole block in two host calls and ten
// launches (model.py:333-343 and its share of loss.backward(), model.py:367).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "../../include/cmhse_hip.h"
#include "gru_ws.hpp"

namespace cmhse {

struct StepLossLayout {
  int32_t R;        // rows of the row-blocked operands: sum over terms of the term's size
  int32_t max_n;    // largest term
  size_t blk_off;   // int32 [n_terms + 1]
  size_t gout;      // float [n_terms]: weight_k * upstream gradient
  size_t y_im, y_s; // float [R, D]: normalised rows, term by term (left / right operand)
  size_t fwd_ws;    // cmhse_contrastive_blocks_fwd's workspace (stored scores first)
  size_t bwd_ws;    // cmhse_contrastive_blocks_bwd's workspace
  size_t d_im, d_s; // float [R, D]
  size_t bytes;
};

// false: the descriptor is not a valid set of terms
static inline bool step_loss_layout(const cmhse_step_losses* d, StepLossLayout* L) {
  if (!d || d->n_emb <= 0 || d->n_emb > CMHSE_STEP_LOSS_MAX || d->n_terms <= 0 ||
      d->n_terms > CMHSE_STEP_LOSS_MAX || d->D <= 0)
    return false;
  int64_t R = 0;
  int32_t mx = 0;
  for (int e = 0; e < d->n_emb; ++e)
    if (d->rows[e] <= 0) return false;
  for (int k = 0; k < d->n_terms; ++k) {
    const int a = d->term_a[k], b = d->term_b[k];
    if (a < 0 || a >= d->n_emb || b < 0 || b >= d->n_emb || d->rows[a] != d->rows[b]) return false;
    R += d->rows[a];
    mx = d->rows[a] > mx ? d->rows[a] : mx;
  }
  if (R > 0x7fffffffLL / d->D) return false;
  L->R = static_cast<int32_t>(R);
  L->max_n = mx;
  const size_t mat = ws_align(static_cast<size_t>(R) * d->D * sizeof(float));
  size_t off = 0;
  L->blk_off = off; off += 256;
  L->gout = off;    off += 256;
  L->y_im = off;    off += mat;
  L->y_s = off;     off += mat;
  L->fwd_ws = off;  off += ws_align(cmhse_contrastive_blocks_workspace(d->n_terms, mx));
  L->bwd_ws = off;  off += ws_align(cmhse_contrastive_blocks_bwd_workspace(d->n_terms, mx));
  L->d_im = off;    off += mat;
  L->d_s = off;     off += mat;
  L->bytes = off;
  return true;
}

}  // namespace cmhse
