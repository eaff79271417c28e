"""Drop-in mirror of the reference's ``src/python/gmm_waymo/src`` EM modules."""
