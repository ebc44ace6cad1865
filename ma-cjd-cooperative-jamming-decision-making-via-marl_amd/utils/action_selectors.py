"""Epsilon-greedy discrete action selection (reference utils/action_selectors.py:4-62).

Same constructor / ``select_action`` signature / ``epsilon`` attribute (read by main.py:259,265).  The
linear schedule is evaluated on the host from ``t_env`` (an int, no device sync).  This class is the
tensor-op form used for host tensors and for callers that bring their own Q-values; on a HIP device
``BasicMAC.select_actions`` normally uses the fused kernel (ops.qhead_select), which applies the same
rule with in-kernel Philox draws."""
import torch


class EpsilonGreedyActionSelector:
    def __init__(self, args):
        self.args = args
        self.epsilon_start = args.epsilon_start
        self.epsilon_finish = args.epsilon_finish
        self.epsilon_anneal_time = args.epsilon_anneal_time
        self.epsilon = self.epsilon_start

    def anneal(self, t_env, test_mode=False):
        """epsilon = max(finish, start - (start - finish) / anneal_time * t_env); frozen in test mode
        (action_selectors.py:30-32)."""
        if not test_mode:
            delta = (self.epsilon_start - self.epsilon_finish) / self.epsilon_anneal_time
            self.epsilon = max(self.epsilon_finish, self.epsilon_start - delta * t_env)
        return self.epsilon

    def select_action(self, agent_qs, avail_actions, t_env, test_mode=False):
        """agent_qs [B, J, A], avail_actions [B, J, A] -> chosen [B, J, 1] (int64)."""
        self.anneal(t_env, test_mode)
        masked_qs = agent_qs.masked_fill(avail_actions == 0, -float("inf"))
        greedy_actions = masked_qs.argmax(dim=2)
        if test_mode:
            return greedy_actions.unsqueeze(-1)
        pick_random = torch.rand_like(agent_qs[:, :, 0]) < self.epsilon
        avail_f = avail_actions.float()
        none_avail = avail_f.sum(dim=-1) == 0  # uniform over all actions if nothing is available
        avail_f = torch.where(none_avail.unsqueeze(-1), torch.full_like(avail_f, 1.0 / avail_f.shape[-1]), avail_f)
        random_actions = torch.multinomial(avail_f.reshape(-1, agent_qs.shape[-1]), num_samples=1) \
            .view(agent_qs.shape[0], agent_qs.shape[1])
        chosen = torch.where(pick_random, random_actions, greedy_actions)
        return chosen.unsqueeze(-1)
