"""Drop-in `vqa` package for the NeuralCX hot path (mirrors the import paths of the reference:
`vqa.models.factory`, `vqa.models.cx.NeuralModel`, `vqa.lib.utils.update_values`)."""
