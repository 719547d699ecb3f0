from .llm_factory import GenericLLMBackbone, get_llm_config
from .nano import (HybridMambaAttentionDynamicCache, NemotronHConfig, NemotronHForCausalLM,
                   NemotronHModel)

__all__ = ["GenericLLMBackbone", "get_llm_config", "NemotronHConfig", "NemotronHForCausalLM",
           "NemotronHModel", "HybridMambaAttentionDynamicCache"]
