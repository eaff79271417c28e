"""Drop-in mirror of the reference's ``src/python/hgmm`` (hierarchical GMM tree + registration)."""
