"""Generic time-loop driver (drop-in for the reference's hmm_layer/BaseRNN.py:149-277).

This is the step-at-a-time plumbing path (BASELINE config 1: a toy cell on CPU): it calls the
cell once per position.  The engine path (MsaHmmLayer) never uses it.  Cells may be
nn.LSTMCell, nn.GRUCell / nn.RNNCell, or any cell whose forward(x_t, state) returns
(output, new_state) with a two-element state (HmmCell, TotalProbabilityCell)."""
import torch
import torch.nn as nn


class BaseRNN(nn.Module):
    def __init__(self, cell, batch_first=False, return_sequences=True, return_state=False, reverse=False):
        super().__init__()
        self.cell = cell
        self.batch_first = batch_first
        self.return_sequences = return_sequences
        self.return_state = return_state
        self.reverse = reverse

    def get_initial_state(self, inputs, batch_size):
        if isinstance(self.cell, nn.LSTMCell):
            z = torch.zeros(batch_size, self.cell.hidden_size, device=inputs.device)
            return (z, z.clone())
        if getattr(self.cell, "get_initial_state", None) is not None:
            return self.cell.get_initial_state(inputs=inputs, batch_size=batch_size)
        return torch.zeros(batch_size, self.cell.hidden_size, device=inputs.device)

    def forward(self, inputs, hidden=None, initial_state=None, **kwargs):
        """inputs (N, T, F) if batch_first else (T, N, F).  ``initial_state`` is an alias of
        ``hidden`` (the reference's layer code passes it under that name, defect D4)."""
        if hidden is None:
            hidden = initial_state
        x = inputs.transpose(0, 1) if self.batch_first else inputs
        if self.reverse:
            x = torch.flip(x, [0])
        steps, batch = x.shape[0], x.shape[1]
        if hidden is None:
            hidden = self.get_initial_state(x, batch)
        lstm = isinstance(self.cell, nn.LSTMCell)
        pair = (not lstm) and isinstance(hidden, (list, tuple)) and len(hidden) == 2
        state = tuple(hidden[:2]) if lstm else hidden
        outs = []
        for t in range(steps):
            if lstm:
                state = self.cell(x[t], state)
                out = state[0]
            elif pair:
                out, state = self.cell(x[t], state)
            else:
                state = self.cell(x[t], state)
                out = state
            outs.append(out)
        if self.return_sequences:
            seq = torch.stack(outs, dim=0)
            if self.batch_first:
                seq = seq.transpose(0, 1)
        else:
            seq = outs[-1].unsqueeze(1) if self.batch_first else outs[-1]
        return (seq, state) if self.return_state else seq
