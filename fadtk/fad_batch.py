from fadtk_amd.fad_batch import cache_embedding_files, _cache_embedding_batch   # noqa: F401
