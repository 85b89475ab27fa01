"""The drop-in boundary on a real MI355X: reference-shaped agents / trainer / env through the
product C-ABI, against the oracle and the golden vectors of the reference's own training loop."""
import pytest

import dropin_checks as DC

pytestmark = pytest.mark.gpu


def test_agent_methods_match_oracle():
    DC.check_agent_methods(H=1, B=1000)
    DC.check_agent_methods(H=0, B=300, seed=4)


def test_adversary_methods_match_oracle():
    DC.check_adversary_methods(B=300)


@pytest.mark.parametrize("name", ["coop_H0", "malicious_H1", "mixed_H1"])
def test_train_RPBCAC_matches_reference_golden(golden, name):
    DC.check_train_golden(golden, name, engine_hook=None)


def test_main_roundtrip_and_warm_start_from_shipped_reference_weights(golden, tmp_path):
    """main.py:52-54,119-121 on the GPU: warm start from the reference's shipped malicious-run weights, oracle parity of the
    continued training, artefact formats, save -> load -> save identity."""
    DC.check_main_roundtrip(golden, tmp_path, None, n_episodes=100, n_ep_fixed=50, max_ep_len=20, n_epochs=3)
