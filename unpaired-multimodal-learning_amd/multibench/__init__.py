"""Host mirror of the reference's MultiBench hot path (MultiBench/models.py, train.py:354-399):
same class names / constructor arguments / state_dict keys, with the per-modality decoder +
masked next-step MSE fused in HIP (umlh_seq_mse_*).  The shared transformer encoder stays on
PyTorch-ROCm ops."""
