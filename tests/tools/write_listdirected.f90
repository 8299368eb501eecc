! write_listdirected.f90 -- test tool: rewrites a one-record-per-line table with LIST-DIRECTED output, the way the
! reference's make_save writes its saves (write(unit,*) ...; SUMMER_SPH.f90:731-735: 9 values per gas particle, 8 per
! sink with the energy written as 0.0).  Under flang list-directed records wrap at 80 columns, so one record spans
! several physical lines -- the ingest of the Fortran hosts has to cope with that (tests/test_host_cpu.py).
!
!   write_listdirected in.txt out.txt ncol        ncol = values per gas row (9: [F] saves, 10: [V] saves)
program write_listdirected
  implicit none
  integer, parameter :: dp = kind(1.0d0)
  character(len=512) :: fin, fout, arg
  character(len=2048) :: line
  real(dp) :: v(10)
  integer :: ncol, ios, k

  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  call get_command_argument(3, arg)
  read(arg, *) ncol
  open(11, file=trim(fin), status='old', action='read')
  open(12, file=trim(fout), status='replace', action='write')
  read(11, '(A)') line
  write(12, *) trim(line)
  do
    read(11, '(A)', iostat=ios) line
    if (ios /= 0) exit
    if (len_trim(line) == 0) cycle
    v = 0.0_dp
    read(line, *, iostat=ios) v(1:ncol)
    if (ios /= 0) read(line, *) v(1:8)           ! a sink row
    if (v(7) /= 0.0_dp) then
      write(12, *) (v(k), k = 1, ncol)
    else
      write(12, *) v(1), v(2), v(3), v(4), v(5), v(6), 0.0, v(8)
    end if
  end do
  close(11)
  close(12)
end program write_listdirected
