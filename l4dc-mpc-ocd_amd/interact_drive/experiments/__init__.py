"""Scenario factories of the reference's experiments/ directory, on the GPU planner."""
