"""Forward + backward wrapper over two sequence layers (drop-in for the reference's
hmm_layer/Bidirectional.py:6-180).  The backward layer receives the time-flipped input and its
output is flipped back, so give it a layer that does NOT flip on its own (SURVEY.md defect D3).
Plumbing path only; the engine computes both directions without materialising flips."""
import torch
import torch.nn as nn


class Bidirectional(nn.Module):
    def __init__(self, layer, backward_layer, merge_mode="concat"):
        super().__init__()
        if not isinstance(layer, nn.Module):
            raise ValueError("`layer` must be an nn.Module, got %r" % (layer,))
        if backward_layer is not None and not isinstance(backward_layer, nn.Module):
            raise ValueError("`backward_layer` must be an nn.Module, got %r" % (backward_layer,))
        if merge_mode not in ("sum", "concat", None):
            raise ValueError("merge_mode must be 'sum', 'concat' or None, got %r" % (merge_mode,))
        for attr in ("batch_first", "hidden_size"):
            if getattr(layer, attr, None) != getattr(backward_layer, attr, None):
                raise ValueError("forward and backward layer differ in %r" % attr)
        self.forward_layer = layer
        self.backward_layer = backward_layer
        self.merge_mode = merge_mode
        self.return_sequences = bool(getattr(layer, "return_sequences", False))
        self.return_state = bool(getattr(layer, "return_state", False))

    def forward(self, sequences, initial_state=None, **kwargs):
        if initial_state is not None:
            half = len(initial_state) // 2
            fstate, bstate = list(initial_state[:half]), list(initial_state[half:])
        else:
            fstate = bstate = None
        axis = 1 if getattr(self.backward_layer, "batch_first", False) else 0
        fres = self.forward_layer(sequences, fstate)
        bres = self.backward_layer(torch.flip(sequences, [axis]), bstate)
        if self.return_state:
            (fout, fstates), (bout, bstates) = fres, bres
        else:
            fout, bout, fstates, bstates = fres, bres, (), ()
        bout = torch.flip(bout, [axis])
        if self.merge_mode == "concat":
            out = torch.cat([fout, bout], dim=-1)
        elif self.merge_mode == "sum":
            out = fout + bout
        else:
            out = [fout, bout]
        if self.return_state:
            return (out, *fstates, *bstates)
        return out
