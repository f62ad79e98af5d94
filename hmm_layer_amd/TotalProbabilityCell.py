"""Chunk-summary cell (drop-in for the reference's hmm_layer/TotalProbabilityCell.py): one
log-space vector x matrix step over the conditional q x q summary of a chunk.  It is the
CPU-visible form of what the engine's scan kernel does in linear space with exponent carry."""
import torch
import torch.nn as nn


class TotalProbabilityCell(nn.Module):
    def __init__(self, cell, reverse=False):
        super().__init__()
        self.cell = cell
        self.reverse = reverse

    @property
    def state_size(self):
        return (torch.Size([self.cell.max_num_states]), torch.Size([]))

    sate_size = state_size            # the reference's spelling

    def make_initial_distribution(self):
        return self.cell.transitioner.make_initial_distribution()

    def forward(self, conditional_forward, states=None, training=None, init=False):
        """conditional_forward (n, q*q) log-probabilities, rows = conditioning state;
        states = (log totals (n, q), _)  ->  (new log totals, (new log totals, loglik))."""
        q = self.cell.max_num_states
        prev, _ = states
        cond = conditional_forward.view(conditional_forward.size(0), q, q)
        total = torch.logsumexp(prev.unsqueeze(-1) + cond, dim=-2)
        return total, (total, torch.logsumexp(total, dim=-1))

    def get_initial_state(self, batch_size=None, inputs=None, dtype=None):
        q = self.cell.max_num_states
        if self.reverse:
            return (torch.zeros(batch_size, q, dtype=dtype), torch.zeros(batch_size, dtype=dtype))
        pi = self.make_initial_distribution()
        pi = pi.repeat(batch_size // self.cell.num_models, 1, 1).transpose(0, 1).reshape(-1, q)
        return (torch.log(pi), torch.zeros(batch_size, dtype=dtype))
