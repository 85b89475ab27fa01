"""The drop-in boundary (reference-shaped Python API) on the hipemu build: CPU-only."""
import pytest

import dropin_checks as DC
from emu_util import emu_lib
from rcmarl_amd import single


@pytest.fixture(autouse=True)
def emu_backend():
    single.set_backend(emu_lib(), "cpu")
    yield
    single._backend = None


def test_agent_methods_match_oracle():
    DC.check_agent_methods(H=1, B=130)


def test_adversary_methods_match_oracle():
    DC.check_adversary_methods(B=70)


def test_grid_world_matches_reference_trajectories(golden):
    DC.check_env_golden(golden)


@pytest.mark.parametrize("name", ["malicious_H1"])
def test_train_RPBCAC_matches_reference_golden(golden, name):
    DC.check_train_golden(golden, name, engine_hook=(emu_lib(), "cpu"))


def test_train_RPBCAC_with_wide_critic_models():
    DC.check_train_wide_critic(engine_hook=(emu_lib(), "cpu"))


def test_main_roundtrip_and_warm_start_from_shipped_reference_weights(golden, tmp_path):
    DC.check_main_roundtrip(golden, tmp_path, (emu_lib(), "cpu"), n_episodes=4, n_ep_fixed=2, max_ep_len=3, n_epochs=1, buffer_size=9)
