from torch import nn


class LoRACompatibleLinear(nn.Linear):
    def forward(self, hidden_states, scale=1.0):
        return super().forward(hidden_states)


class LoRACompatibleConv(nn.Conv2d):
    def forward(self, hidden_states, scale=1.0):
        return super().forward(hidden_states)
