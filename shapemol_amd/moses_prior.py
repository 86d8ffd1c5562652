"""Pooled atom-count histogram of the MOSES training/validation set.

Derived from the reference's data blob data/MOSES2_training_val_shape_atomnum_dict.pkl
(1531 voxel-size keys, 150 000 molecules; consumed at scripts/sample_diffusion.py:218,245-253)
by summing the per-voxel-size histograms.  Mean 21.38 atoms, range 9..27.
"""
ATOM_NUMS = (9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27)
ATOM_FREQ = (11, 24, 27, 55, 103, 415, 591, 1589, 4267, 8996, 17013, 19540, 21830, 23113,
             22492, 18944, 9094, 1895, 1)
