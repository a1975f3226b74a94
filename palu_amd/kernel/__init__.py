"""Host-side mirror of the reference's `kernel/` package for the decode path (`abx_rope.abx`,
`palu_attention.LlamaPaluAttention` / `HeadwiseLowRankModule`) -- same names, arguments and errors, HIP underneath."""
