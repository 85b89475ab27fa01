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
