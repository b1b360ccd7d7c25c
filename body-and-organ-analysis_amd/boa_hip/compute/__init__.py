"""Mirror of the reference's `body_organ_analysis.compute` function surface for the hot path (same names,
argument meaning and error behaviour); see INTEGRATION.md for the one-line swap in BOA/commands.py."""
