"""Cars of the reference's reward-inference tests (interact_drive/reward_design/tests)."""
