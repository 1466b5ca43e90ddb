"""Mirror of the reference's `acr` package surface for the hot path (model, result_parser, mano_wrapper, main, utils)."""
