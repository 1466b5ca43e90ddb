"""Mirror of the reference's `mano` package surface (manolayer.ManoLayer)."""
