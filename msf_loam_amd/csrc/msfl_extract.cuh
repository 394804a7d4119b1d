// msfl_extract.cuh — feature-extraction kernels (stage A). Filled in below.
#pragma once
