"""Module names of the reference's Cython islands (util/cython/*), now doors onto the HIP engine."""
