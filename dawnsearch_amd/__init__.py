"""dawnsearch_amd — MI355X-native embed-and-rank hot path of DawnSearch (src/embedding + src/search).

The product is `libdawn_hip.so` (hand-written HIP for gfx950 behind the C ABI of include/dawn_hip.h).
This package is the host-side mirror of the reference's provider interfaces over that ABI.  Importing it
without the built shared object raises ImportError — there is no CPU or torch fallback.
"""
from ._lib import (DawnError, NotNormalizedError, EM_LEN, MAX_K, LIB_PATH, device_count, last_error)  # noqa: F401
from .index import (VectorIndex, BestResults, is_normalized, normalize, to24, from24, topk_merge_device,
                    topk_merge_packed_device, result_blob_bytes)  # noqa: F401
from .search_provider import (SearchProvider, SearchResult, FoundPage, ExtractedPage, SearchStats,  # noqa: F401
                              search_remote_merge)
from .embedding_provider import EmbeddingProvider, write_synthetic_model  # noqa: F401,E402
from .tokenizer import Tokenizer  # noqa: F401,E402
from .sharded import ShardedSearch, shard_range, merge_host  # noqa: F401,E402
