"""The 15-state gene model's A and pi for the timing scripts, from the package's own transitioner
(the oracle is test infrastructure and is not imported outside tests/)."""
import torch

from hmm_layer_amd.gene_pred_hmm_transitioner import GenePredMultiHMMTransitioner


def gene15(device):
    tr = GenePredMultiHMMTransitioner(initial_exon_len=200, initial_intron_len=4500, initial_ir_len=10000,
                                      starting_distribution_init="zeros").to(device)
    with torch.no_grad():
        return tr.make_A().contiguous(), tr.make_initial_distribution().reshape(1, -1).contiguous()
