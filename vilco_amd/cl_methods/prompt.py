"""L2P prompt pool used by the ViLCo method inside `forward` (reference:
MQ/libs/cl_methods/prompt.py:4-116): cosine similarity between the mean text token and learnable
keys, top-k (or a fixed per-task window during training) prompts of `length` tokens each are
prepended to the text sequence.  [B, L, 768] tensors with B = 2: plain device tensor ops."""
import torch
import torch.nn as nn


class Prompt(nn.Module):
    def __init__(self, length=5, embed_dim=768, embedding_key='mean', prompt_init='uniform', prompt_pool=False,
                 prompt_key=False, pool_size=None, top_k=None, batchwise_prompt=False, prompt_key_init='uniform'):
        super().__init__()
        self.length, self.embed_dim, self.prompt_pool = length, embed_dim, prompt_pool
        self.embedding_key, self.prompt_init = embedding_key, prompt_init
        self.pool_size, self.top_k, self.batchwise_prompt = pool_size, top_k, batchwise_prompt
        if not prompt_pool:
            raise NotImplementedError("the MQ configs only build Prompt with prompt_pool=True (meta_archs.py:643)")

        def make(shape, init):
            p = nn.Parameter(torch.zeros(shape) if init == 'zero' else torch.randn(shape))
            if init == 'uniform':
                nn.init.uniform_(p, -1, 1)
            return p
        self.prompt = make((pool_size, length, embed_dim), prompt_init)
        if prompt_key:
            self.prompt_key = make((pool_size, embed_dim), prompt_key_init)
        else:
            self.prompt_key = torch.mean(self.prompt, dim=1)

    @staticmethod
    def l2_normalize(x, dim=None, epsilon=1e-12):
        sq = torch.sum(x ** 2, dim=dim, keepdim=True)
        return x * torch.rsqrt(torch.clamp_min(sq, epsilon))       # == maximum(sq, eps): no host scalar to upload

    def forward(self, x_embed, prompt_mask=None, cls_features=None):
        if self.embedding_key == 'mean':
            x_key = torch.mean(x_embed, dim=1)
        elif self.embedding_key == 'max':
            x_key = torch.max(x_embed, dim=1)[0]
        elif self.embedding_key == 'mean_max':
            x_key = torch.max(x_embed, dim=1)[0] + 2 * torch.mean(x_embed, dim=1)
        elif self.embedding_key == 'cls':
            x_key = torch.max(x_embed, dim=1)[0] if cls_features is None else cls_features
        else:
            raise NotImplementedError("Not supported way of calculating embedding keys!")
        prompt_norm = self.l2_normalize(self.prompt_key, dim=1)
        x_norm = self.l2_normalize(x_key, dim=1)
        similarity = x_norm @ prompt_norm.t()                       # [B, pool]

        if prompt_mask is None:
            _, idx = torch.topk(similarity, k=self.top_k, dim=1)
            if self.batchwise_prompt:
                ids, counts = torch.unique(idx, return_counts=True, sorted=True)
                if ids.shape[0] < self.pool_size:
                    pad = self.pool_size - ids.shape[0]
                    ids = torch.cat([ids, torch.full((pad,), torch.min(idx.flatten()), device=ids.device)])
                    counts = torch.cat([counts, torch.full((pad,), 0, device=counts.device)])
                _, major = torch.topk(counts, k=self.top_k)
                idx = ids[major].expand(x_embed.shape[0], -1)
        else:
            idx = prompt_mask

        raw = self.prompt[idx]                                      # [B, top_k, length, C]
        B, k, length, c = raw.shape
        batched_prompt = raw.reshape(B, k * length, c)
        key_sel = prompt_norm[idx]                                  # [B, top_k, C]
        reduce_sim = torch.sum(key_sel * x_norm.unsqueeze(1)) / x_embed.shape[0]
        return {'prompt_idx': idx, 'prompt_norm': prompt_norm, 'x_embed_norm': x_norm,
                'similarity': similarity, 'selected_key': key_sel, 'reduce_sim': reduce_sim,
                'total_prompt_len': batched_prompt.shape[1],
                'prompted_embedding': torch.cat([batched_prompt, x_embed], dim=1)}
