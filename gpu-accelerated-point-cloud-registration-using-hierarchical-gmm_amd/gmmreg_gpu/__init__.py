"""Drop-in mirror of the reference's ``src/python/gmmreg_gpu`` EM modules."""
