"""Mirror of the reference package layout (consumers of the hot-path kernels only)."""
