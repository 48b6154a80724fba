"""Fitting API: defs (constants), simple (driver functions), expert (ExpertSolver)."""
