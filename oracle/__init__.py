"""CPU oracle for the HMM forward / backward / posterior / Viterbi hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``hmm_layer_amd/`` may import this
package: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / the timed CPU baseline,
never as the thing that is shipped or measured on the GPU side.

Contents
--------
``textbook``  float64 numpy restatement of the scaled forward-backward with the
              reference's epsilon clamps (accuracy yardstick at every size).
``ref_cell``  op-for-op PyTorch-CPU fp32 restatement of the reference cell step
              and its drivers (the "reference CPU path" timed as cpu_baseline):
              hmm_layer/MsaHmmCell.py:73-142, hmm_layer/TotalProbabilityCell.py:30-63,
              hmm_layer/MsaHMMLayer.py:227-521, hmm_layer/BaseRNN.py:194-248.
``params``    numpy restatement of the gene-prediction transitioner / emitter /
              k-mer producers (hmm_layer/gene_pred_hmm_transitioner.py,
              hmm_layer/gene_pred_hmm_emitter.py, hmm_layer/kmer.py).
``viterbi``   max-plus Viterbi in Q-format fixed point (no reference
              implementation exists: parity unpinned).
``hmm_oracle.c``  plain-C twin of ``textbook``/``viterbi`` for the large sizes.

Parity pinning: ``ref_cell`` and ``params`` are pinned bit-for-bit / to 1 ulp
against fixtures captured from the imported reference
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``) and against the
k-mer outputs recorded in the reference's ``tests/test_tf.ipynb``.
Viterbi: parity unpinned (the reference has no Viterbi).
"""
