"""Mirror of the reference package layout (hot-path modules only)."""
