"""CPU: host-side logic of the boundary - config composition, parameter trees, metrics helpers."""
import numpy as np
import pytest
import torch

from mava_amd.config import check_total_timesteps, compose


def test_compose_defaults_match_reference_surface():
    c = compose("default_ff_mappo")
    s = c.system
    assert (c.arch.num_envs, c.arch.num_evaluation, c.arch.num_eval_episodes) == (16, 200, 32)
    assert (s.num_updates, s.seed, s.add_agent_id, s.update_batch_size, s.rollout_length) == (1000, 42, True, 2, 128)
    assert (s.actor_lr, s.critic_lr, s.ppo_epochs, s.num_minibatches) == (2.5e-4, 2.5e-4, 4, 2)
    assert (s.gamma, s.gae_lambda, s.clip_eps, s.ent_coef, s.vf_coef, s.max_grad_norm) == (0.99, 0.95, 0.2, 0.01, 0.5, 0.5)
    assert s.decay_learning_rates is False and s.total_timesteps is None
    assert c.network.actor_network.pre_torso.layer_sizes == [128, 128]
    assert c.env.scenario.task_name == "tiny-2ag" and c.env.kwargs.time_limit == 500
    assert compose("default_rec_mappo").system.recurrent_chunk_size is None
    assert compose("default_rec_ippo").network.hidden_state_dim == 128


def test_overrides():
    c = compose("default_ff_ippo", ["env/scenario=tiny-4ag", "arch.num_envs=4096", "system.update_batch_size=1",
                                    "system.total_timesteps=~", "system.decay_learning_rates=true", "+system.foo=3"])
    assert c.env.scenario.task_config.num_agents == 4 and c.arch.num_envs == 4096 and c.system.foo == 3
    assert c.system.decay_learning_rates is True
    c.system.num_agents = 4  # struct mode off: new keys allowed (ff_mappo.py:560, :341)
    assert c.system.num_agents == 4
    with pytest.raises(ValueError):
        compose("default_ff_isac")
    # network group swap like the reference's `network=continuous_mlp` (configs/network/continuous_mlp.yaml)
    from mava_amd.networks import ContinuousActionHead, DiscreteActionHead, make_action_head

    assert isinstance(make_action_head(compose("default_ff_mappo", ["network=continuous_mlp"]).network.action_head, 3),
                      ContinuousActionHead)
    assert isinstance(make_action_head(c.network.action_head, 5), DiscreteActionHead)
    assert ContinuousActionHead(3, independent_std=False).independent_std is False  # (runs on the general network path)
    assert ContinuousActionHead(3, min_scale=1e-2).min_scale == 1e-2  # (a kernel argument since round 3: networks.py:134,162)
    with pytest.raises(ValueError):
        ContinuousActionHead(3, min_scale=-1.0)
    # network group swap to the CNN torsos (configs/network/cnn.yaml) and the general path's selection rule
    from mava_amd.generic_networks import CNNTorso, is_default_mlp, torso_from_config

    cnn = compose("default_ff_ippo", ["network=cnn"]).network
    assert isinstance(torso_from_config(cnn.actor_network.pre_torso), CNNTorso) and not is_default_mlp(cnn.actor_network.pre_torso)
    assert is_default_mlp(c.network.actor_network.pre_torso)
    ln = compose("default_ff_ippo", ["network.actor_network.pre_torso.use_layer_norm=true"]).network
    assert not is_default_mlp(ln.actor_network.pre_torso)


def test_total_timesteps():
    c = check_total_timesteps(compose("default_ff_mappo"), n_devices=1)
    assert c.system.total_timesteps == 1 * 1000 * 128 * 2 * 16
    c = compose("default_ff_mappo", ["system.total_timesteps=20000000"])
    check_total_timesteps(c, n_devices=8)
    assert c.system.num_updates == 20000000 // 128 // 2 // 16 // 8


def test_param_tree_is_a_view_of_the_flat_buffer():
    from mava_amd.networks import DiscreteActionHead, FeedForwardActor, FeedForwardValueNet, MLPTorso

    actor = FeedForwardActor(MLPTorso([128, 128]), DiscreteActionHead(5), 70)
    critic = FeedForwardValueNet(MLPTorso([128, 128]), True, 264)
    assert (actor.num_params, critic.num_params) == (26245, 50561)
    flat = actor.init_flat(3)
    tree = actor.tree(flat, lead=(1, 2))
    k = tree["params"]["torso"]["Dense_0"]["kernel"]
    assert k.shape == (1, 2, 70, 128) and k.data_ptr() == flat.data_ptr()
    assert tree["params"]["action_head"]["Dense_0"]["bias"].shape == (1, 2, 5)
    w = flat[:70 * 128].view(70, 128).double()
    assert torch.allclose(w @ w.T, 2.0 * torch.eye(70, dtype=torch.float64), atol=1e-5)  # orthogonal(sqrt 2), rows<cols
    head = tree["params"]["action_head"]["Dense_0"]["kernel"][0, 0].double()
    assert torch.allclose(head.T @ head, 1e-4 * torch.eye(5, dtype=torch.float64), atol=1e-8)  # orthogonal(0.01)
    assert float(tree["params"]["torso"]["Dense_1"]["bias"].abs().max()) == 0.0
    back = actor.flat_from_tree(tree)
    assert torch.equal(back, flat)
    ctree = critic.tree(critic.init_flat(4))
    assert set(ctree["params"]) == {"torso", "Dense_0"} and ctree["params"]["Dense_0"]["kernel"].shape == (128, 1)
    with pytest.raises(NotImplementedError):
        MLPTorso([64, 64])
    with pytest.raises(NotImplementedError):
        FeedForwardActor(MLPTorso([128]), DiscreteActionHead(5), 70)


def test_continuous_head_param_trees():
    """ContinuousActionHead (networks.py:127-169): Flax names action_head/{mean, log_std}; log_std (zeros) follows the
    network in the flat vector; tree <-> flat round trips for the feed-forward and the recurrent actor."""
    from mava_amd.networks import ContinuousActionHead, FeedForwardActor, MLPTorso
    from mava_amd.rec_networks import RecurrentActor, rec_segments

    ff = FeedForwardActor(MLPTorso([128, 128]), ContinuousActionHead(3), 31)
    assert ff.continuous and ff.num_params == ff.num_mlp_params + 3 == 31 * 128 + 128 + 128 * 128 + 128 + 128 * 3 + 3 + 3
    flat = ff.init_flat(5)
    assert float(flat[-3:].abs().max()) == 0.0  # nn.initializers.zeros
    flat[-3:] = torch.tensor([0.1, -0.2, 0.3])
    tree = ff.tree(flat, lead=(1, 2))
    ah = tree["params"]["action_head"]
    assert set(ah) == {"mean", "log_std"} and ah["mean"]["kernel"].shape == (1, 2, 128, 3) and ah["log_std"].shape == (1, 2, 3)
    assert ah["log_std"].data_ptr() == flat[-3:].data_ptr()
    assert torch.equal(ff.flat_from_tree(tree), flat)

    rec = RecurrentActor(MLPTorso([128]), MLPTorso([128]), ContinuousActionHead(2), 20)
    n_net = rec_segments(20, 2)[1]
    assert rec.continuous and rec.num_params == n_net + 2
    rflat = rec.init_flat(6)
    rflat[-2:] = torch.tensor([0.5, -0.5])
    rtree = rec.tree(rflat, lead=(1, 1))
    assert set(rtree["params"]["action_head"]) == {"mean", "log_std"}
    assert torch.equal(rec.flat_from_tree(rtree), rflat)
    assert torch.equal(rec.log_std(rflat), rflat[-2:])


def test_final_step_metrics():
    from mava_amd.learner import get_final_step_metrics

    m = {"episode_return": torch.arange(6.0).view(1, 1, 1, 2, 3), "episode_length": torch.arange(6).view(1, 1, 1, 2, 3),
         "is_terminal_step": torch.tensor([0, 1, 0, 0, 0, 1]).view(1, 1, 1, 2, 3).bool()}
    fm, done = get_final_step_metrics(m)
    assert done and fm["episode_return"].tolist() == [1.0, 5.0]
    m["is_terminal_step"] = torch.zeros_like(m["is_terminal_step"])
    fm, done = get_final_step_metrics(m)
    assert not done and float(fm["episode_return"].sum()) == 0.0


def test_categorical_host_view():
    from mava_amd.distributions import Categorical

    logits = torch.tensor([[0.0, 1.0, 2.0], [3.0, 0.0, 0.0]])
    mask = torch.tensor([[True, True, False], [True, True, True]])
    d = Categorical(logits, mask)
    assert d.mode().tolist() == [1, 0]
    lp = d.log_prob(torch.tensor([1, 0]))
    assert np.isclose(float(lp[0]), 1.0 - np.log(np.exp(0) + np.exp(1)))
    assert float(d.entropy()[0]) > 0
    s = d.sample(seed=torch.Generator().manual_seed(0))
    assert int(s[0]) in (0, 1)


def test_tanh_normal_host_view():
    """The evaluator-seam distribution of the continuous head against the oracle formulas (distributions.py:24-91)."""
    from mava_amd.distributions import TanhNormal
    from oracle import tanh_normal as tn

    rng = np.random.default_rng(0)
    loc, ls = rng.normal(size=(6, 3)), rng.normal(size=3) * 0.3
    d = TanhNormal(torch.from_numpy(loc), torch.from_numpy(ls))
    assert torch.allclose(d.mode(), torch.tanh(torch.from_numpy(loc)))
    a = np.tanh(loc + 0.2)
    a[0, 0], a[1, 1] = 0.9999, -1.0
    np.testing.assert_allclose(d.log_prob(torch.from_numpy(a)).numpy(), tn.log_prob(a, loc, ls), rtol=1e-6, atol=1e-6)
    s = d.sample(seed=torch.Generator().manual_seed(0))
    assert s.shape == (6, 3) and float(s.abs().max()) < 1.0 and torch.isfinite(d.entropy(seed=torch.Generator().manual_seed(1))).all()


def test_side_stream_probe_without_gpu():
    """mava_amd/streams.py: no side stream off the GPU (the learners then keep everything on the one stream)."""
    from mava_amd.streams import overlapping_stream

    assert overlapping_stream("cpu") is None
