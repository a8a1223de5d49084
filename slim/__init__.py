"""Drop-in module names of the reference package: `slim.prune`, `slim.lora`, `slim.eval`, `slim.utils`,
`slim.layerwrapper`, `slim.quantization.quantization`, `slim.quantization.utils`, `slim.data` resolve to
the slim_amd implementations."""
